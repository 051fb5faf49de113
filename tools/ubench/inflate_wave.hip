/*
 * tools/ubench/inflate_wave.hip -- GATE for a device-side inflate of FASTQ .gz ("next" item 3; would replace
 * gzread under vendor/kseq.h:229 with compressed bytes crossing PCIe instead of text).
 *
 * Question: how many GB/s of TEXT can one MI355X inflate when every wave decodes one chunk of a deflate stream on its own?
 * The host pool (host/inflate_spec.cpp + gz_parallel.cpp) does 9.0 GB/s on 16 CPUs; the go / no-go line is 40 GB/s.
 *
 * DEFLATE decoding is serial in two ways: (1) the Huffman symbols -- each code's length decides where the next one starts --
 * and (2) the LZ77 copies -- each may read what the previous ones wrote.  The gate is staged accordingly:
 *   stage 1 (this kernel, inflate_tokens)  one wave per chunk, Huffman tables in LDS, the bit reader and the table walk executed
 *            wave-uniformly (scalar registers; the 64 lanes serve as the input buffer, v_readlane, and as the token output buffer): block headers, dynamic / fixed / stored blocks -> a stream of 32-bit tokens (literal | length, distance).
 *            No window, no copies: it is an UPPER BOUND for any complete decoder, at a fraction of its cost.
 *   stage 2  (only if stage 1 clears the line) the LZ77 expansion in the marker format of host/inflate_spec.cpp + window resolve.
 * Every chunk's tokens are verified on the host against the original text (literal bytes, every copied byte, lengths).
 *
 * Input: a synthetic FASTQ with the bench's realistic quality model (synth.h model 1), compressed here with zlib level 6 as ONE
 * raw deflate stream (pieces of 16 MiB of text deflated in parallel and joined by sync flushes, as pigz writes them).
 * Block boundaries come from zlib itself (inflate with Z_BLOCK): a chunk = the blocks from the first boundary at or behind
 * chunk_bytes * i to the next chunk's start.
 *
 *   hipcc -O3 --offload-arch=gfx950 -o build/ubench/inflate_wave tools/ubench/inflate_wave.hip ntsm_amd/csrc/synth_host.cpp -lz -pthread
 *   build/ubench/inflate_wave [--reads 4e6] [--chunk-kib 128] [--level 6] [--threads 16]
 */
#include <hip/hip_runtime.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include "../../include/ntsm_synth.h"

#define HIPOK(call)                                                                                       \
	do {                                                                                                  \
		hipError_t e_ = (call);                                                                           \
		if (e_ != hipSuccess) { fprintf(stderr, "inflate_wave: %s: %s\n", #call, hipGetErrorString(e_)); exit(1); } \
	} while (0)

struct ChunkDesc {
	unsigned long long start_bit, end_bit;      /* bit offsets into the raw deflate stream: [first block's header, end of the last block) */
	unsigned long long tok_off;                 /* this chunk's slice of the token buffer */
	unsigned long long text_off, text_len;      /* where its output lies in the text (host-side verification) */
};

/* ---- device ------------------------------------------------------------------------------------------------------------------ */
constexpr int kLitRoot = 10, kDistRoot = 8, kPreRoot = 7;

struct WaveLds {
	uint32_t lit[1 << kLitRoot];                /* entry: bits 0-3 code length (0 = longer than the root: slow path), 4-7 extra bits, 8-9 kind, 16-31 value */
	uint32_t dist[1 << kDistRoot];
	uint32_t pre[1 << kPreRoot];
	uint32_t lit_cnt[16], dist_cnt[16], pre_cnt[16];      /* codes per length */
	uint32_t offs[16], first[16];
	uint16_t lit_sorted[320], dist_sorted[32], pre_sorted[32];   /* symbols by (length, symbol): canonical order */
	uint8_t lens[352];
};

enum { KIND_LIT = 0, KIND_BASE = 1, KIND_EOB = 2, KIND_BAD = 3 };

__device__ __forceinline__ uint32_t lit_entry(uint32_t sym, uint32_t l)
{
	if (sym < 256) return l | (KIND_LIT << 8) | (sym << 16);
	if (sym == 256) return l | (KIND_EOB << 8);
	const uint32_t i = sym - 257;
	if (i > 28) return l | (KIND_BAD << 8);
	uint32_t xb = 0, base;
	if (i < 8) base = 3 + i;
	else if (i == 28) base = 258;
	else { xb = (i - 4) >> 2; base = 3 + ((4 + (i & 3)) << xb); }
	return l | (xb << 4) | (KIND_BASE << 8) | (base << 16);
}
__device__ __forceinline__ uint32_t dist_entry(uint32_t sym, uint32_t l)
{
	if (sym > 29) return l | (KIND_BAD << 8);
	uint32_t xb = 0, base;
	if (sym < 4) base = 1 + sym;
	else { xb = (sym >> 1) - 1; base = 1 + ((2 + (sym & 1)) << xb); }
	return l | (xb << 4) | (KIND_BASE << 8) | (base << 16);
}
__device__ __forceinline__ uint32_t pre_entry(uint32_t sym, uint32_t l) { return l | (sym << 16); }

/* Canonical Huffman table from code lengths, built by the whole wave: root table of 2^ROOT entries for codes up to ROOT bits,
 * `sorted` + `cnt` for the bit-at-a-time walk of the longer ones (rare by construction: each has probability < 2^-ROOT). */
template <int ROOT, class Mk>
__device__ void build_table(const uint8_t *lens, int n_syms, uint32_t *table, uint16_t *sorted, uint32_t *cnt, uint32_t *offs, uint32_t *first, Mk mk, int lane)
{
	if (lane < 16) cnt[lane] = 0;
	for (int i = lane; i < (1 << ROOT); i += 64) table[i] = 0;
	__syncthreads();
	for (int s = lane; s < n_syms; s += 64) {
		const uint32_t l = lens[s];
		if (l) atomicAdd(&cnt[l], 1u);
	}
	__syncthreads();
	if (lane == 0) {
		uint32_t o = 0, code = 0;
		cnt[0] = 0;
		for (int l = 1; l <= 15; ++l) {
			code = (code + cnt[l - 1]) << 1;
			first[l] = code;
			offs[l] = o;
			o += cnt[l];
		}
	}
	__syncthreads();
	uint32_t run[16];
#pragma unroll
	for (int l = 0; l < 16; ++l) run[l] = 0;
	for (int base = 0; base < n_syms; base += 64) {
		const int s = base + lane;
		const uint32_t l = s < n_syms ? lens[s] : 0u;
		uint32_t rank = 0;
#pragma unroll
		for (int L = 1; L <= 15; ++L) {
			const unsigned long long m = __ballot(l == (uint32_t) L);
			if (l == (uint32_t) L) rank = run[L] + (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
			run[L] += (uint32_t) __popcll(m);
		}
		if (l) {
			sorted[offs[l] + rank] = (uint16_t) s;
			if (l <= (uint32_t) ROOT) {
				const uint32_t code = first[l] + rank;
				const uint32_t rev = __brev(code) >> (32 - l);
				const uint32_t e = mk((uint32_t) s, l);
				for (uint32_t i = rev; i < (1u << ROOT); i += 1u << l) table[i] = e;
			}
		}
	}
	__syncthreads();
}

#define UNI(x) ((uint32_t) __builtin_amdgcn_readfirstlane((int) (x)))

struct BitReader {
	unsigned long long bb;
	uint32_t bc, ip;                            /* bits in bb; index of the next dword to take */
	uint32_t vin, vnext;                        /* per lane: dwords [ip & ~63, +64) and the 64 behind them */
	const uint32_t *comp;
	int lane;
	__device__ __forceinline__ void init(const uint32_t *c, unsigned long long start_bit, int ln)
	{
		comp = c; lane = ln;
		ip = (uint32_t) (start_bit >> 5);
		const uint32_t b0 = ip & ~63u;
		vin = comp[b0 + lane];
		vnext = comp[b0 + 64 + lane];
		bb = 0; bc = 0;
		refill();
		const uint32_t drop = (uint32_t) (start_bit & 31);
		bb >>= drop; bc -= drop;
		if (bc < 32) refill();
	}
	__device__ __forceinline__ void refill()      /* bc < 32 -> one more dword */
	{
		const uint32_t d = (uint32_t) __builtin_amdgcn_readlane((int) vin, (int) (ip & 63u));
		bb |= (unsigned long long) d << bc;
		bc += 32;
		++ip;
		if ((ip & 63u) == 0) {
			vin = vnext;
			vnext = comp[ip + 64 + lane];
		}
	}
	__device__ __forceinline__ void need() { if (bc < 32) refill(); }
	__device__ __forceinline__ uint32_t take(uint32_t n)
	{
		const uint32_t v = (uint32_t) bb & ((1u << n) - 1u);
		bb >>= n; bc -= n;
		return v;
	}
	__device__ __forceinline__ unsigned long long pos() const { return (unsigned long long) ip * 32ull - bc; }
};

/* the walk for a code longer than the root table: one bit at a time against the canonical counts (RFC 1951 3.2.2) */
__device__ __forceinline__ uint32_t slow_symbol(BitReader &br, const uint32_t *cnt, const uint16_t *sorted, uint32_t *len_out)
{
	uint32_t code = 0, first = 0, index = 0;
	for (uint32_t len = 1; len <= 15; ++len) {
		code |= br.take(1);
		const uint32_t count = UNI(cnt[len]);
		if (code - first < count) { *len_out = len; return UNI(sorted[index + (code - first)]); }
		index += count;
		first = (first + count) << 1;
		code <<= 1;
	}
	*len_out = 0;
	return 0xFFFFu;
}

__constant__ uint8_t kPreOrder[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };

#ifndef INFLATE_WAVES_PER_SIMD
#define INFLATE_WAVES_PER_SIMD 4               /* register budget: 4 = what the compiler takes by itself (100 VGPRs); -DINFLATE_WAVES_PER_SIMD=8 caps it at 64 */
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(INFLATE_WAVES_PER_SIMD, INFLATE_WAVES_PER_SIMD))) void inflate_tokens(const uint32_t *__restrict__ comp, const ChunkDesc *__restrict__ chunks, uint32_t *__restrict__ tokens,
		unsigned long long *__restrict__ n_tokens, unsigned long long *__restrict__ n_out, uint32_t *__restrict__ status, uint32_t chunk_base)
{
	__shared__ WaveLds L;
	const int lane = threadIdx.x;
	const uint32_t me = blockIdx.x + chunk_base;             /* a launch may cover a segment of the chunks (stage 2 overlaps segments) */
	const ChunkDesc cd = chunks[me];
	BitReader br;
	br.init(comp, cd.start_bit, lane);
	uint32_t *const tok = tokens + cd.tok_off;
	uint32_t tokbuf = 0;
	unsigned long long ntok = 0, nout = 0;
	uint32_t err = 0, blocks = 0;
	auto emit = [&](uint32_t t) {
		tokbuf = (uint32_t) lane == ((uint32_t) ntok & 63u) ? t : tokbuf;      /* v_cmp + v_cndmask with two scalar operands: a v_writelane */
		++ntok;
		if (((uint32_t) ntok & 63u) == 0) tok[ntok - 64 + lane] = tokbuf;
	};
	while (br.pos() < cd.end_bit && !err) {
		br.need();
		(void) br.take(1);                                      /* BFINAL: the chunk ends where the host says */
		const uint32_t btype = br.take(2);
		++blocks;
		if (btype == 0) {                                       /* stored */
			(void) br.take(br.bc & 7u);
			br.need();
			const uint32_t len = br.take(16);
			br.need();
			const uint32_t nlen = br.take(16);
			if ((len ^ nlen) != 0xFFFFu) { err = 1; break; }
			for (uint32_t i = 0; i < len; ++i) { br.need(); emit(br.take(8)); }
			nout += len;
			continue;
		}
		if (btype == 3) { err = 2; break; }
		uint32_t hlit, hdist;
		if (btype == 1) {                                       /* fixed code */
			hlit = 288; hdist = 30;
			for (int s = lane; s < 320; s += 64) L.lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5;
			__syncthreads();
		} else {
			hlit = br.take(5) + 257;
			hdist = br.take(5) + 1;
			const uint32_t hclen = br.take(4) + 4;
			if (lane < 19) L.lens[lane] = 0;
			__syncthreads();
			for (uint32_t i = 0; i < hclen; ++i) {
				br.need();
				const uint32_t v = br.take(3);
				if (lane == 0) L.lens[kPreOrder[i]] = (uint8_t) v;
			}
			__syncthreads();
			build_table<kPreRoot>(L.lens, 19, L.pre, L.pre_sorted, L.pre_cnt, L.offs, L.first, pre_entry, lane);
			const uint32_t n = hlit + hdist;
			uint32_t i = 0, prev = 0;
			while (i < n) {
				br.need();
				const uint32_t e = UNI(L.pre[(uint32_t) br.bb & ((1u << kPreRoot) - 1u)]);
				if ((e & 15u) == 0) { err = 3; break; }
				(void) br.take(e & 15u);
				const uint32_t sym = e >> 16;
				uint32_t rep = 1, val = sym;
				if (sym == 16) { rep = 3 + br.take(2); val = prev; }
				else if (sym == 17) { rep = 3 + br.take(3); val = 0; }
				else if (sym == 18) { rep = 11 + br.take(7); val = 0; }
				if (i + rep > n) { err = 4; break; }
				for (uint32_t j = (uint32_t) lane; j < rep; j += 64) L.lens[i + j] = (uint8_t) val;
				i += rep;
				prev = val;
			}
			if (err) break;
			__syncthreads();
		}
		build_table<kLitRoot>(L.lens, (int) hlit, L.lit, L.lit_sorted, L.lit_cnt, L.offs, L.first, lit_entry, lane);
		build_table<kDistRoot>(L.lens + hlit, (int) hdist, L.dist, L.dist_sorted, L.dist_cnt, L.offs, L.first, dist_entry, lane);
		/* ---- the symbols of the block */
		for (;;) {
			br.need();
			uint32_t e = UNI(L.lit[(uint32_t) br.bb & ((1u << kLitRoot) - 1u)]);
			if ((e & 15u) == 0) {
				uint32_t l;
				const uint32_t sym = slow_symbol(br, L.lit_cnt, L.lit_sorted, &l);
				if (!l) { err = 5; break; }
				e = lit_entry(sym, l);
			} else (void) br.take(e & 15u);
			const uint32_t kind = (e >> 8) & 3u;
			if (kind == KIND_LIT) { emit(e >> 16); ++nout; continue; }
			if (kind == KIND_EOB) break;
			if (kind == KIND_BAD) { err = 6; break; }
			const uint32_t len = (e >> 16) + br.take((e >> 4) & 15u);
			br.need();
			uint32_t d = UNI(L.dist[(uint32_t) br.bb & ((1u << kDistRoot) - 1u)]);
			if ((d & 15u) == 0) {
				uint32_t l;
				const uint32_t sym = slow_symbol(br, L.dist_cnt, L.dist_sorted, &l);
				if (!l) { err = 7; break; }
				d = dist_entry(sym, l);
			} else (void) br.take(d & 15u);
			if (((d >> 8) & 3u) != KIND_BASE) { err = 8; break; }
			const uint32_t dd = (d >> 16) + br.take((d >> 4) & 15u);
			emit(0x80000000u | (len << 16) | (dd - 1u));
			nout += len;
		}
	}
	if ((uint32_t) ntok & 63u) {                                /* the last, partial group of tokens */
		if ((uint32_t) lane < ((uint32_t) ntok & 63u)) tok[(ntok & ~63ull) + lane] = tokbuf;
	}
	if (lane == 0) {
		n_tokens[me] = ntok;
		n_out[me] = nout;
		status[me] = err | (br.pos() == cd.end_bit ? 0u : 0x100u) | (blocks << 16);
	}
}

/* ---- stage 2: LZ77 expansion of the tokens into 16-bit marker symbols (the format of host/inflate_spec.cpp) ------------------------
 * One wave per EXPANSION chunk = a group of consecutive Huffman chunks (the groups bound the length of the window chain below).  The
 * wave knows nothing of the 32 KiB in front of its group: a copy that reaches there produces MARKER | index-into-that-window, and
 * markers propagate through later copies like any symbol.  Tokens are taken 64 at a time (one per lane): a prefix sum of the lengths
 * places them; literals and matches whose source lies entirely in front of the batch are copied by their own lanes in parallel --
 * from an LDS ring that holds the last kRing symbols, or from global memory when the source is older than the ring (73 % of FASTQ's
 * matches) --, matches that read the batch's own output follow one after the other, each copied by the whole wave (lane = symbol,
 * source index modulo the distance for self-overlapping runs).  The batch is then flushed from the ring with coalesced stores. */
constexpr uint32_t kMarker = 0x8000u, kWindow = 32768u;
constexpr uint32_t kRing = 8192u, kBatchCap = 2048u;     /* symbols in the LDS ring; most a batch may produce */

struct XChunk { uint32_t first_chunk, n_chunks; unsigned long long out_off, out_len; };

__global__ __launch_bounds__(64) void inflate_expand(const uint32_t *__restrict__ tokens, const ChunkDesc *__restrict__ chunks, const unsigned long long *__restrict__ n_tokens,
		const XChunk *__restrict__ xchunks, uint16_t *sym, uint32_t *__restrict__ xstatus, uint16_t *__restrict__ tails, uint32_t x_base)
{
	__shared__ uint16_t ring[kRing];
	const int lane = threadIdx.x;
	const uint32_t me = blockIdx.x + x_base;
	const XChunk xc = xchunks[me];
	uint16_t *const tail = tails + (unsigned long long) me * kWindow;          /* this chunk's last 32 Ki symbols once more, 64 KiB-aligned, for the window chain */
	const long long tail_from = (long long) xc.out_len - (long long) kWindow;
	uint16_t *const out = sym + xc.out_off;
	long long P = 0;                                         /* symbols produced so far (position inside this expansion chunk) */
	uint32_t err = 0;
	for (uint32_t c = xc.first_chunk; c < xc.first_chunk + xc.n_chunks; ++c) {
		const uint32_t *tok = tokens + chunks[c].tok_off;
		const unsigned long long nt = n_tokens[c];
		for (unsigned long long b = 0; b < nt;) {
			const bool have = b + lane < nt;
			const uint32_t t = have ? tok[b + lane] : 0u;
			const bool is_match = have && (t >> 31);
			uint32_t len = is_match ? (t >> 16) & 0x1FFu : (have ? 1u : 0u);
			const uint32_t dist = is_match ? (t & 0x7FFFu) + 1u : 0u;
			uint32_t incl = len;                              /* inclusive prefix sum over the lanes */
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
			/* a batch produces at most kBatchCap symbols: the lanes behind that wait for the next one */
			const unsigned long long fit = __ballot(have && incl <= kBatchCap);
			const uint32_t k = (uint32_t) __popcll(fit);      /* >= 1: a token is at most 258 symbols */
			const bool mine = (uint32_t) lane < k;
			if (!mine) len = 0;
			const uint32_t N = (uint32_t) __shfl(incl, (int) k - 1, 64);
			const uint32_t start = incl - (mine ? len : 0u);
			const long long s0 = P + (long long) start - (long long) dist;   /* first source position of a match */
			const bool external = mine && is_match && s0 + (long long) len <= P;
			const unsigned long long dep = __ballot(mine && is_match && !external);
			/* The far reads below must see this wave's own flushes of earlier batches: the stores have left the wave once vmcnt is 0
			 * (they are write-through to the L2) and the loads are agent-scope atomic loads, which do not take a stale line from the
			 * L1.  (A __threadfence() here -- L2 write-back + invalidate, thousands of them per millisecond chip-wide -- held the
			 * whole kernel to 10 GB/s whatever the number of waves.) */
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			if (mine && !is_match) ring[(uint32_t) (P + start) & (kRing - 1u)] = (uint16_t) (t & 0xFFu);
			{   /* external matches: every lane copies its own, eight symbols per step so that eight far loads are in flight */
				const uint32_t elen = external ? len : 0u;
				uint32_t maxlen = elen;
#pragma unroll
				for (int o = 32; o > 0; o >>= 1) maxlen = max(maxlen, (uint32_t) __shfl_xor(maxlen, o, 64));
				for (uint32_t i0 = 0; i0 < maxlen; i0 += 8) {
					uint16_t v[8];
#pragma unroll
					for (uint32_t j = 0; j < 8; ++j) {
						const uint32_t i = i0 + j;
						v[j] = 0;
						if (i < elen) {
							const long long sp = s0 + i;
							if (sp < 0) v[j] = (uint16_t) (kMarker | (uint32_t) ((long long) kWindow + sp));
							else if (sp >= P - (long long) (kRing - kBatchCap)) v[j] = ring[(uint32_t) sp & (kRing - 1u)];
							else v[j] = __hip_atomic_load(out + sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						}
					}
#pragma unroll
					for (uint32_t j = 0; j < 8; ++j) {
						const uint32_t i = i0 + j;
						if (i < elen) ring[(uint32_t) (P + start + i) & (kRing - 1u)] = v[j];
					}
				}
			}
			__syncthreads();
			for (unsigned long long d = dep; d; d &= d - 1) {   /* matches that read this batch's own output: in token order, the wave copies each */
				const int t0 = __builtin_ctzll(d);
				const uint32_t L = (uint32_t) __shfl(len, t0, 64), S = (uint32_t) __shfl(start, t0, 64), D = (uint32_t) __shfl(dist, t0, 64);
				const long long base = P + (long long) S - (long long) D;
				for (uint32_t i0 = 0; i0 < L; i0 += 64) {
					const uint32_t i = i0 + (uint32_t) lane;
					uint16_t v = 0;
					if (i < L) {
						const long long sp = base + (long long) (D < L ? i % D : i);   /* D, L are wave-uniform: the division runs for self-overlapping matches only */
						v = sp < 0 ? (uint16_t) (kMarker | (uint32_t) ((long long) kWindow + sp)) : ring[(uint32_t) sp & (kRing - 1u)];
					}
					__syncthreads();
					if (i < L) ring[(uint32_t) (P + S + i) & (kRing - 1u)] = v;
				}
				__syncthreads();
			}
			for (uint32_t j = (uint32_t) lane; j < N; j += 64) {
				const uint16_t v = ring[(uint32_t) (P + j) & (kRing - 1u)];
				out[P + j] = v;
				if (P + j >= tail_from) tail[P + j - tail_from] = v;
			}
			P += N;
			b += k;
			if (k == 0) { err = 1; break; }
		}
		if (err) break;
	}
	if (lane == 0) xstatus[me] = err | ((unsigned long long) P == xc.out_len ? 0u : 2u);
}

/* The window chain: window(x) = the last 32 KiB of expansion chunk x - 1 as BYTES, resolved against window(x - 1).  Serial over the
 * expansion chunks (each step needs the one before), 32,768 look-ups per step by one workgroup with both windows in LDS. */
__global__ __launch_bounds__(1024) void inflate_chain(const uint16_t *__restrict__ tails, uint32_t x0, uint32_t x1, uint8_t *__restrict__ windows)
{
	__shared__ __attribute__((aligned(16))) uint8_t w[2][kWindow];
	const int t = threadIdx.x;                               /* thread t owns window bytes [32 t, 32 t + 32) = 64 bytes of symbols */
	{   /* window(x0 - 1) from the launch that ended there (all zero in front of the first chunk: nothing refers to it) */
		uint4 *dst = reinterpret_cast<uint4 *>(w[(x0 - 1) & 1] + 32 * t);
		if (x0 > 1) {
			const uint4 *src = reinterpret_cast<const uint4 *>(windows + (unsigned long long) (x0 - 1) * kWindow + 32 * t);
			dst[0] = src[0]; dst[1] = src[1];
		} else dst[0] = dst[1] = make_uint4(0, 0, 0, 0);
	}
	uint4 q[4];
	{
		const uint4 *tl = reinterpret_cast<const uint4 *>(tails + (unsigned long long) (x0 - 1) * kWindow + 32 * t);
		q[0] = tl[0]; q[1] = tl[1]; q[2] = tl[2]; q[3] = tl[3];
	}
	__syncthreads();
	for (uint32_t x = x0; x < x1; ++x) {                     /* window(x) = tail(x - 1) resolved against window(x - 1) */
		const uint8_t *prev = w[(x - 1) & 1];
		uint8_t *cur = w[x & 1];
		const uint32_t sw[16] = { q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w, q[2].x, q[2].y, q[2].z, q[2].w, q[3].x, q[3].y, q[3].z, q[3].w };
		if (x + 1 < x1) {                                    /* the next tail does not depend on the chain: it is on its way while this one resolves */
			const uint4 *tl = reinterpret_cast<const uint4 *>(tails + (unsigned long long) x * kWindow + 32 * t);
			q[0] = tl[0]; q[1] = tl[1]; q[2] = tl[2]; q[3] = tl[3];
		}
		uint32_t packed[8];
#pragma unroll
		for (int i = 0; i < 32; ++i) {
			const uint32_t sv = (sw[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
			const uint32_t v = (sv & kMarker) ? prev[sv & (kWindow - 1u)] : (sv & 0xFFu);
			if ((i & 3) == 0) packed[i >> 2] = v; else packed[i >> 2] |= v << (8 * (i & 3));
		}
		uint4 *cw = reinterpret_cast<uint4 *>(cur + 32 * t);
		uint4 *gw = reinterpret_cast<uint4 *>(windows + (unsigned long long) x * kWindow + 32 * t);
		const uint4 lo = make_uint4(packed[0], packed[1], packed[2], packed[3]), hi = make_uint4(packed[4], packed[5], packed[6], packed[7]);
		cw[0] = lo; cw[1] = hi;
		gw[0] = lo; gw[1] = hi;
		__syncthreads();
	}
}

/* Resolve: every symbol to its byte -- a literal is itself, a marker the byte of its expansion chunk's window. */
__global__ __launch_bounds__(256) void inflate_resolve(const uint16_t *__restrict__ sym, const XChunk *__restrict__ xchunks, const uint8_t *__restrict__ windows, uint8_t *__restrict__ text, uint32_t x_base)
{
	const XChunk xc = xchunks[blockIdx.y + x_base];
	const uint8_t *w = windows + (unsigned long long) (blockIdx.y + x_base) * kWindow;
	for (unsigned long long i = (unsigned long long) blockIdx.x * 256 + threadIdx.x; i < xc.out_len; i += (unsigned long long) gridDim.x * 256) {
		const uint16_t s = sym[xc.out_off + i];
		text[xc.out_off + i] = (s & kMarker) ? w[s & (kWindow - 1u)] : (uint8_t) s;
	}
}

/* ---- host -------------------------------------------------------------------------------------------------------------------- */
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void parallel(unsigned n, const std::function<void(unsigned)> &f)
{
	std::vector<std::thread> th;
	for (unsigned t = 1; t < n; ++t) th.emplace_back(f, t);
	f(0);
	for (auto &x : th) x.join();
}

struct Boundary { uint64_t bit, out; };

int main(int argc, char **argv)
{
	uint64_t n_reads = 4000000, chunk_kib = 128, piece_mib = 16, group = 8;
	uint32_t segments = 8;
	int level = 6, reps = 3;
	unsigned threads = 16;
	std::string fastq;
	for (int i = 1; i + 1 < argc; i += 2) {
		const std::string k = argv[i], v = argv[i + 1];
		if (k == "--reads") n_reads = (uint64_t) strtod(v.c_str(), nullptr);
		else if (k == "--chunk-kib") chunk_kib = strtoull(v.c_str(), nullptr, 10);
		else if (k == "--level") level = atoi(v.c_str());
		else if (k == "--threads") threads = (unsigned) atoi(v.c_str());
		else if (k == "--reps") reps = atoi(v.c_str());
		else if (k == "--fastq") fastq = v;
		else if (k == "--segments") segments = std::max(1, atoi(v.c_str()));
		else if (k == "--group") group = strtoull(v.c_str(), nullptr, 10);   /* Huffman chunks per expansion chunk (stage 2) */
		else { fprintf(stderr, "inflate_wave: unknown option %s\n", k.c_str()); return 2; }
	}
	/* ---- the text */
	std::vector<uint8_t> text;
	{
		std::string path = fastq;
		if (path.empty()) {
			char tmp[] = "/tmp/inflate_wave_XXXXXX";
			const int fd = mkstemp(tmp);
			if (fd < 0) { perror("mkstemp"); return 1; }
			close(fd);
			path = tmp;
			const uint32_t n_sites = 96287;
			std::vector<uint8_t> win((size_t) n_sites * 2 * NTSM_SYNTH_WSTRIDE);
			if (ntsm_synth_sites(20241218ull, n_sites, 19, win.data(), nullptr, nullptr)) return 1;
			ntsm_synth_short sp;
			ntsm_synth_short_params(&sp, 7, 150, n_sites, 0.10, 0.01, 0.0005);
			if (ntsm_synth_short_write_fastq_mt_q(&sp, win.data(), 0, n_reads, path.c_str(), threads, 1)) return 1;
		}
		FILE *f = fopen(path.c_str(), "rb");
		if (!f) { perror(path.c_str()); return 1; }
		fseek(f, 0, SEEK_END);
		text.resize((size_t) ftell(f));
		fseek(f, 0, SEEK_SET);
		if (fread(text.data(), 1, text.size(), f) != text.size()) return 1;
		fclose(f);
		if (fastq.empty()) unlink(path.c_str());
	}
	const uint64_t n_text = text.size();
	/* ---- one raw deflate stream, pieces deflated in parallel and joined by sync flushes */
	const uint64_t piece = piece_mib << 20, n_pieces = (n_text + piece - 1) / piece;
	std::vector<std::vector<uint8_t>> cp(n_pieces);
	double t0 = now_s();
	{
		std::atomic<uint64_t> next(0);
		parallel(threads, [&](unsigned) {
			for (;;) {
				const uint64_t i = next++;
				if (i >= n_pieces) return;
				const uint64_t off = i * piece, len = std::min(piece, n_text - off);
				z_stream z;
				memset(&z, 0, sizeof z);
				deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
				cp[i].resize(deflateBound(&z, len) + 64);
				z.next_in = text.data() + off; z.avail_in = (uInt) len;
				z.next_out = cp[i].data(); z.avail_out = (uInt) cp[i].size();
				deflate(&z, i + 1 == n_pieces ? Z_FINISH : Z_SYNC_FLUSH);
				cp[i].resize(z.total_out);
				deflateEnd(&z);
			}
		});
	}
	const double deflate_s = now_s() - t0;
	std::vector<uint64_t> cp_off(n_pieces + 1, 0);
	for (uint64_t i = 0; i < n_pieces; ++i) cp_off[i + 1] = cp_off[i] + cp[i].size();
	const uint64_t n_comp = cp_off[n_pieces];
	std::vector<uint8_t> comp(n_comp + 4096, 0);
	for (uint64_t i = 0; i < n_pieces; ++i) memcpy(comp.data() + cp_off[i], cp[i].data(), cp[i].size());
	/* ---- block boundaries from zlib (Z_BLOCK), piece by piece in parallel; and zlib's own single-thread inflate rate */
	std::vector<std::vector<Boundary>> pb(n_pieces);
	t0 = now_s();
	{
		std::atomic<uint64_t> next(0);
		parallel(threads, [&](unsigned) {
			std::vector<uint8_t> sink(1 << 16);
			for (;;) {
				const uint64_t i = next++;
				if (i >= n_pieces) return;
				z_stream z;
				memset(&z, 0, sizeof z);
				inflateInit2(&z, -15);
				z.next_in = cp[i].data(); z.avail_in = (uInt) cp[i].size();
				pb[i].push_back({ cp_off[i] * 8, i * piece });
				for (;;) {
					z.next_out = sink.data(); z.avail_out = (uInt) sink.size();
					const int rc = inflate(&z, Z_BLOCK);
					if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) { fprintf(stderr, "inflate_wave: zlib rc %d\n", rc); exit(1); }
					if (z.data_type & 128)                               /* just behind a block's end (bit 64 only says "in the last block") */
						pb[i].push_back({ (cp_off[i] + z.total_in) * 8 - (uint64_t) (z.data_type & 63), i * piece + z.total_out });
					if (rc == Z_STREAM_END || (z.avail_in == 0 && z.avail_out != 0)) break;
				}
				inflateEnd(&z);
			}
		});
	}
	const double zlib_s = now_s() - t0;
	for (auto &v : cp) { v.clear(); v.shrink_to_fit(); }
	std::vector<Boundary> bnd;
	for (uint64_t i = 0; i < n_pieces; ++i)
		for (auto &b : pb[i]) if (bnd.empty() || b.bit > bnd.back().bit) bnd.push_back(b);
	if (bnd.back().out != n_text) { fprintf(stderr, "inflate_wave: boundaries end at %llu of %llu text bytes\n", (unsigned long long) bnd.back().out, (unsigned long long) n_text); return 1; }
	/* ---- chunks */
	std::vector<ChunkDesc> chunks;
	{
		const uint64_t cbits = chunk_kib * 1024 * 8;
		size_t j = 0;
		uint64_t tok_off = 0;
		while (j + 1 < bnd.size()) {
			const uint64_t target = (bnd[j].bit / cbits + 1) * cbits;
			size_t e = j + 1;
			while (e + 1 < bnd.size() && bnd[e].bit < target) ++e;
			ChunkDesc c;
			c.start_bit = bnd[j].bit; c.end_bit = bnd[e].bit;
			c.text_off = bnd[j].out; c.text_len = bnd[e].out - bnd[j].out;
			c.tok_off = tok_off;
			tok_off += (c.text_len + 63) & ~63ull;              /* a token produces at least one byte */
			chunks.push_back(c);
			j = e;
		}
		fprintf(stderr, "inflate_wave: %.2f GB text, %.2f GB deflate (ratio %.2f, level %d, %.1f s on %u threads), %zu blocks (%.0f KB text each), %zu chunks of ~%llu KiB; "
			"zlib inflate + Z_BLOCK: %.2f s on %u threads\n", n_text / 1e9, n_comp / 1e9, (double) n_text / n_comp, level, deflate_s, threads, bnd.size() - 1,
			n_text / 1e3 / (bnd.size() - 1), chunks.size(), (unsigned long long) chunk_kib, zlib_s, threads);
	}
	const uint64_t tok_total = chunks.back().tok_off + ((chunks.back().text_len + 63) & ~63ull);
	/* ---- device */
	uint32_t *d_comp, *d_tok, *d_status;
	ChunkDesc *d_chunks;
	unsigned long long *d_ntok, *d_nout;
	HIPOK(hipMalloc((void **) &d_comp, comp.size()));
	HIPOK(hipMalloc((void **) &d_tok, tok_total * 4));
	HIPOK(hipMalloc((void **) &d_chunks, chunks.size() * sizeof(ChunkDesc)));
	HIPOK(hipMalloc((void **) &d_ntok, chunks.size() * 8));
	HIPOK(hipMalloc((void **) &d_nout, chunks.size() * 8));
	HIPOK(hipMalloc((void **) &d_status, chunks.size() * 4));
	HIPOK(hipMemcpy(d_comp, comp.data(), comp.size(), hipMemcpyHostToDevice));
	HIPOK(hipMemcpy(d_chunks, chunks.data(), chunks.size() * sizeof(ChunkDesc), hipMemcpyHostToDevice));
	hipEvent_t ea, eb;
	HIPOK(hipEventCreate(&ea));
	HIPOK(hipEventCreate(&eb));
	float best_ms = 1e30f;
	for (int r = 0; r < reps + 1; ++r) {
		HIPOK(hipEventRecord(ea, 0));
		hipLaunchKernelGGL(inflate_tokens, dim3((unsigned) chunks.size()), dim3(64), 0, 0, d_comp, d_chunks, d_tok, d_ntok, d_nout, d_status, 0u);
		HIPOK(hipEventRecord(eb, 0));
		HIPOK(hipEventSynchronize(eb));
		HIPOK(hipGetLastError());
		float ms;
		HIPOK(hipEventElapsedTime(&ms, ea, eb));
		if (r) best_ms = std::min(best_ms, ms);
		fprintf(stderr, "inflate_wave: launch %d: %.3f ms\n", r, ms);
	}
	/* ---- verification: every token of every chunk against the text */
	std::vector<unsigned long long> ntok(chunks.size()), nout(chunks.size());
	std::vector<uint32_t> status(chunks.size());
	HIPOK(hipMemcpy(ntok.data(), d_ntok, chunks.size() * 8, hipMemcpyDeviceToHost));
	HIPOK(hipMemcpy(nout.data(), d_nout, chunks.size() * 8, hipMemcpyDeviceToHost));
	HIPOK(hipMemcpy(status.data(), d_status, chunks.size() * 4, hipMemcpyDeviceToHost));
	std::vector<uint32_t> tok(tok_total);
	HIPOK(hipMemcpy(tok.data(), d_tok, tok_total * 4, hipMemcpyDeviceToHost));
	std::atomic<uint64_t> bad(0), lits(0), matches(0), match_bytes(0), far(0);
	{
		std::atomic<uint64_t> next(0);
		parallel(threads, [&](unsigned) {
			uint64_t l = 0, m = 0, mb = 0, fr = 0, b = 0;
			for (;;) {
				const uint64_t c = next++;
				if (c >= chunks.size()) break;
				const ChunkDesc &cd = chunks[c];
				if ((status[c] & 0xFFFF) || nout[c] != cd.text_len) { ++b; continue; }
				uint64_t pos = cd.text_off;
				const uint32_t *t = tok.data() + cd.tok_off;
				bool ok = true;
				for (uint64_t i = 0; i < ntok[c] && ok; ++i) {
					if (!(t[i] >> 31)) { ok = text[pos] == (uint8_t) t[i] && t[i] < 256; ++pos; ++l; }
					else {
						const uint32_t len = (t[i] >> 16) & 0x1FF, dist = (t[i] & 0x7FFF) + 1;
						if (dist > pos || pos + len > cd.text_off + cd.text_len) { ok = false; break; }
						ok = memcmp(text.data() + pos, text.data() + pos - dist, std::min<uint32_t>(len, dist)) == 0;
						for (uint32_t j = dist; j < len && ok; ++j) ok = text[pos + j] == text[pos + j - dist];
						pos += len; ++m; mb += len;
						if (dist > 2048) ++fr;
					}
				}
				if (!ok || pos != cd.text_off + cd.text_len) ++b;
			}
			lits += l; matches += m; match_bytes += mb; far += fr; bad += b;
		});
	}
	uint64_t total_tok = 0, nblocks = 0;
	for (size_t c = 0; c < chunks.size(); ++c) { total_tok += ntok[c]; nblocks += status[c] >> 16; }
	/* ---- stage 2: tokens -> 16-bit marker symbols -> window chain -> bytes; the bytes must be the text */
	std::vector<XChunk> xch;
	for (size_t c = 0; c < chunks.size();) {
		XChunk x;
		x.first_chunk = (uint32_t) c; x.n_chunks = 0; x.out_off = chunks[c].text_off; x.out_len = 0;
		while (c < chunks.size() && (x.n_chunks < group || x.out_len <= kWindow)) { x.out_len += chunks[c].text_len; ++x.n_chunks; ++c; }
		xch.push_back(x);
	}
	if (xch.size() > 1 && xch.back().out_len <= kWindow) {         /* a short last group joins the one before it */
		XChunk last = xch.back(); xch.pop_back();
		xch.back().n_chunks += last.n_chunks; xch.back().out_len += last.out_len;
	}
	uint16_t *d_sym, *d_tails; uint8_t *d_win, *d_text; XChunk *d_x; uint32_t *d_xst;
	HIPOK(hipMalloc((void **) &d_sym, n_text * 2 + 64));
	HIPOK(hipMalloc((void **) &d_tails, xch.size() * (size_t) kWindow * 2));
	HIPOK(hipMalloc((void **) &d_win, xch.size() * (size_t) kWindow));
	HIPOK(hipMalloc((void **) &d_text, n_text + 64));
	HIPOK(hipMalloc((void **) &d_x, xch.size() * sizeof(XChunk)));
	HIPOK(hipMalloc((void **) &d_xst, xch.size() * 4));
	HIPOK(hipMemcpy(d_x, xch.data(), xch.size() * sizeof(XChunk), hipMemcpyHostToDevice));
	float ms_tok = 0, ms_expand = 1e30f, ms_chain = 1e30f, ms_resolve = 1e30f, ms_all = 1e30f, ms_overlap = 1e30f;
	hipEvent_t e0, e1, e2, e3, e4;
	HIPOK(hipEventCreate(&e0)); HIPOK(hipEventCreate(&e1)); HIPOK(hipEventCreate(&e2)); HIPOK(hipEventCreate(&e3)); HIPOK(hipEventCreate(&e4));
	const uint32_t nx = (uint32_t) xch.size();
	/* (a) the four kernels one after the other on one stream: what each phase costs */
	for (int r = 0; r < reps + 1; ++r) {
		HIPOK(hipMemsetAsync(d_sym, 0xEE, n_text * 2, 0));
		HIPOK(hipMemsetAsync(d_text, 0, n_text, 0));
		HIPOK(hipEventRecord(e0, 0));
		hipLaunchKernelGGL(inflate_tokens, dim3((unsigned) chunks.size()), dim3(64), 0, 0, d_comp, d_chunks, d_tok, d_ntok, d_nout, d_status, 0u);
		HIPOK(hipEventRecord(e1, 0));
		hipLaunchKernelGGL(inflate_expand, dim3(nx), dim3(64), 0, 0, d_tok, d_chunks, d_ntok, d_x, d_sym, d_xst, d_tails, 0u);
		HIPOK(hipEventRecord(e2, 0));
		if (nx > 1) hipLaunchKernelGGL(inflate_chain, dim3(1), dim3(1024), 0, 0, d_tails, 1u, nx, d_win);
		HIPOK(hipEventRecord(e3, 0));
		hipLaunchKernelGGL(inflate_resolve, dim3(64, nx), dim3(256), 0, 0, d_sym, d_x, d_win, d_text, 0u);
		HIPOK(hipEventRecord(e4, 0));
		HIPOK(hipEventSynchronize(e4));
		HIPOK(hipGetLastError());
		float a, b2, c2, d2, all;
		HIPOK(hipEventElapsedTime(&a, e0, e1)); HIPOK(hipEventElapsedTime(&b2, e1, e2)); HIPOK(hipEventElapsedTime(&c2, e2, e3));
		HIPOK(hipEventElapsedTime(&d2, e3, e4)); HIPOK(hipEventElapsedTime(&all, e0, e4));
		fprintf(stderr, "inflate_wave: complete decode %d: tokens %.3f + expand %.3f + chain %.3f + resolve %.3f = %.3f ms\n", r, a, b2, c2, d2, all);
		if (r && all < ms_all) { ms_all = all; ms_tok = a; ms_expand = b2; ms_chain = c2; ms_resolve = d2; }
	}
	/* (b) overlapped: the file in `segments` pieces; the token kernels (scalar-unit bound) run back to back on one stream, the
	 * expansion / chain / resolve of a piece (vector, LDS and memory work) follow on a second stream as soon as its tokens exist */
	{
		hipStream_t sa, sb;
		HIPOK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
		HIPOK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
		std::vector<hipEvent_t> tok_done(segments);
		for (auto &e : tok_done) HIPOK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
		for (int r = 0; r < reps + 1; ++r) {
			HIPOK(hipMemset(d_sym, 0xEE, n_text * 2));
			HIPOK(hipMemset(d_text, 0, n_text));
			HIPOK(hipDeviceSynchronize());
			const double t0 = now_s();
			for (uint32_t sg = 0; sg < segments; ++sg) {
				const uint32_t xa = (uint32_t) ((uint64_t) nx * sg / segments), xb = (uint32_t) ((uint64_t) nx * (sg + 1) / segments);
				if (xa == xb) { HIPOK(hipEventRecord(tok_done[sg], sa)); continue; }
				const uint32_t ca = xch[xa].first_chunk, cb = xb < nx ? xch[xb].first_chunk : (uint32_t) chunks.size();
				hipLaunchKernelGGL(inflate_tokens, dim3(cb - ca), dim3(64), 0, sa, d_comp, d_chunks, d_tok, d_ntok, d_nout, d_status, ca);
				HIPOK(hipEventRecord(tok_done[sg], sa));
				HIPOK(hipStreamWaitEvent(sb, tok_done[sg], 0));
				hipLaunchKernelGGL(inflate_expand, dim3(xb - xa), dim3(64), 0, sb, d_tok, d_chunks, d_ntok, d_x, d_sym, d_xst, d_tails, xa);
				if (xb > std::max(xa, 1u)) hipLaunchKernelGGL(inflate_chain, dim3(1), dim3(1024), 0, sb, d_tails, std::max(xa, 1u), xb, d_win);
				hipLaunchKernelGGL(inflate_resolve, dim3(64, xb - xa), dim3(256), 0, sb, d_sym, d_x, d_win, d_text, xa);
			}
			HIPOK(hipStreamSynchronize(sa));
			HIPOK(hipStreamSynchronize(sb));
			HIPOK(hipGetLastError());
			const float ms = (float) ((now_s() - t0) * 1e3);
			fprintf(stderr, "inflate_wave: complete decode, %u segments on two streams, %d: %.3f ms\n", segments, r, ms);
			if (r) ms_overlap = std::min(ms_overlap, ms);
		}
	}
	uint64_t x_bad = 0, first_diff = ~0ull;
	{
		std::vector<uint32_t> xst(xch.size());
		HIPOK(hipMemcpy(xst.data(), d_xst, xch.size() * 4, hipMemcpyDeviceToHost));
		for (uint32_t v : xst) x_bad += v != 0;
		std::vector<uint8_t> got(n_text);
		HIPOK(hipMemcpy(got.data(), d_text, n_text, hipMemcpyDeviceToHost));
		if (memcmp(got.data(), text.data(), n_text) != 0)
			for (uint64_t i = 0; i < n_text; ++i) if (got[i] != text[i]) { first_diff = i; break; }
	}
	const bool full_ok = !x_bad && first_diff == ~0ull && !bad.load();
	if (!full_ok) fprintf(stderr, "inflate_wave: COMPLETE DECODE WRONG: %llu expansion chunks flagged, first differing byte %lld\n", (unsigned long long) x_bad, (long long) first_diff);
	const double s = best_ms / 1e3;
	printf("{\"complete\": {\"what\": \"tokens -> LZ77 expansion into 16-bit marker symbols (one wave per group of %llu chunks) -> window chain -> resolve to bytes\", "
		"\"expansion_chunks\": %zu, \"one_after_the_other\": {\"tokens_ms\": %.3f, \"expand_ms\": %.3f, \"chain_ms\": %.3f, \"resolve_ms\": %.3f, \"total_ms\": %.3f, \"text_GBps\": %.2f}, "
		"\"overlapped\": {\"segments\": %u, \"streams\": 2, \"total_ms\": %.3f, \"text_GBps\": %.2f}, \"bytes_equal_the_text\": %s, \"gate_GBps\": 40.0}, ",
		(unsigned long long) group, xch.size(), ms_tok, ms_expand, ms_chain, ms_resolve, ms_all, n_text / (ms_all / 1e3) / 1e9,
		segments, ms_overlap, n_text / (ms_overlap / 1e3) / 1e9, full_ok ? "true" : "false");
	printf("\"stage\": 1, \"what\": \"Huffman decode to tokens, one wave per chunk (no LZ77 copies: an upper bound for a complete decoder)\", "
		"\"text_bytes\": %llu, \"deflate_bytes\": %llu, \"ratio\": %.3f, \"level\": %d, \"chunks\": %zu, \"chunk_KiB\": %llu, \"blocks\": %llu, "
		"\"kernel_ms\": %.3f, \"text_GBps\": %.2f, \"deflate_GBps\": %.2f, \"tokens\": %llu, \"Gtokens_per_s\": %.2f, \"bytes_per_token\": %.3f, "
		"\"literals\": %llu, \"matches\": %llu, \"avg_match_len\": %.2f, \"matches_farther_than_2048\": %.3f, \"lds_bytes_per_wave\": %zu, "
		"\"chunks_wrong\": %llu, \"verified\": %s, \"zlib_inflate_s_on_%u_threads\": %.3f, \"gate_GBps\": 40.0}\n",
		(unsigned long long) n_text, (unsigned long long) n_comp, (double) n_text / n_comp, level, chunks.size(), (unsigned long long) chunk_kib, (unsigned long long) nblocks,
		best_ms, n_text / s / 1e9, n_comp / s / 1e9, (unsigned long long) total_tok, total_tok / s / 1e9, (double) n_text / total_tok,
		(unsigned long long) lits.load(), (unsigned long long) matches.load(), matches.load() ? (double) match_bytes.load() / matches.load() : 0.0,
		matches.load() ? (double) far.load() / matches.load() : 0.0, sizeof(WaveLds), (unsigned long long) bad.load(), bad.load() ? "false" : "true", threads, zlib_s);
	return bad.load() || !full_ok ? 1 : 0;
}
