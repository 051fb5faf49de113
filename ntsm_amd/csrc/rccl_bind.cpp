/*
 * rccl_bind.cpp -- the one collective of the count path: SUM all-reduce of every context's dense count vector + totals
 * (uint64[n_kmers + 4]) across the GPUs of ONE process (ntsm_allreduce; `ntsmCount -g a,b`), RCCL over xGMI; contexts that
 * share a device are summed on that device first.
 * SUM, not MAX: the per-site maxima are taken on the host from the summed per-k-mer counts, which is what a single
 * reference run computes (src/FingerPrint.hpp:281-294); the reference itself has no multi-device path (only
 * `omp parallel for` over files on one shared table, src/FingerPrint.hpp:46-48).
 * One rank per GPU under torch.distributed does the same reduction from Python (ntsm_amd/dist.py).
 */
#include <dlfcn.h>
#include <rccl/rccl.h>                     /* types only: the library is bound with dlopen on first use */

#include <mutex>
#include <utility>
#include <vector>

#include "ntsm_internal.h"

namespace ntsm_rt {

namespace {

/* RCCL is bound on first use: the library is 570 MB of code objects that the HIP runtime would otherwise map and
 * register in every process that links it (~0.1 s and 1.4 GB of RSS for a single-GPU ntsmCount). */
struct Rccl {
	decltype(&ncclCommInitAll) CommInitAll = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclAllReduce) AllReduce = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	bool ok = false;
	Rccl()
	{
		void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
		if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
		if (!h) return;
		CommInitAll = (decltype(CommInitAll)) dlsym(h, "ncclCommInitAll");
		GroupStart = (decltype(GroupStart)) dlsym(h, "ncclGroupStart");
		GroupEnd = (decltype(GroupEnd)) dlsym(h, "ncclGroupEnd");
		AllReduce = (decltype(AllReduce)) dlsym(h, "ncclAllReduce");
		CommDestroy = (decltype(CommDestroy)) dlsym(h, "ncclCommDestroy");
		ok = CommInitAll && GroupStart && GroupEnd && AllReduce && CommDestroy;
	}
};
const Rccl &rccl_bind()
{
	static Rccl rccl;                                     /* thread-safe one-time binding */
	return rccl;
}

} // namespace

bool rccl_available() { return rccl_bind().ok; }

/* RCCL SUM between contexts on DISTINCT devices (one communicator rank per device). */
static int rccl_between_devices(ntsm_ctx *const *ctxs, int n)
{
	const Rccl &rccl = rccl_bind();
	if (!rccl.ok) return NTSM_ERR_RCCL;
	std::vector<int> devs(n);
	for (int i = 0; i < n; ++i) devs[i] = ctxs[i]->device;
	/* Communicators are kept per device list for the life of the process: ncclCommInitAll costs far more than the
	 * 12 MB reduction it serves (tens of milliseconds against well under one). */
	static std::mutex comm_mu;
	static std::vector<std::pair<std::vector<int>, std::vector<ncclComm_t>>> comm_cache;
	std::lock_guard<std::mutex> comm_lock(comm_mu);
	std::vector<ncclComm_t> *comms = nullptr;
	for (auto &e : comm_cache) if (e.first == devs) comms = &e.second;
	if (!comms) {
		std::vector<ncclComm_t> made(n);
		if (rccl.CommInitAll(made.data(), n, devs.data()) != ncclSuccess) return NTSM_ERR_RCCL;
		comm_cache.emplace_back(devs, made);
		comms = &comm_cache.back().second;
	}
	bool ok = rccl.GroupStart() == ncclSuccess;
	for (int i = 0; i < n && ok; ++i) {
		ok = hipSetDevice(devs[i]) == hipSuccess &&
			rccl.AllReduce(ctxs[i]->d_vec, ctxs[i]->d_vec, (size_t) ctxs[i]->n_kmers + 4, ncclUint64, ncclSum,
					(*comms)[i], ctxs[i]->rstream) == ncclSuccess;
	}
	ok = (rccl.GroupEnd() == ncclSuccess) && ok;
	for (int i = 0; i < n; ++i) {
		(void) hipSetDevice(devs[i]);
		if (hipStreamSynchronize(ctxs[i]->rstream) != hipSuccess) ok = false;
	}
	if (!ok) return NTSM_ERR_RCCL;
	return NTSM_OK;
}

/* The merge of ntsm_allreduce.  Contexts that share a device need no collective: their vectors are summed into the first
 * of them (through the host: 8 bytes per site k-mer, once per run), RCCL then runs between one context per distinct device, and
 * the result is copied back to the others on the device.  With a single distinct device -- `ntsmCount -g 0,0`, two contexts
 * on one GPU -- RCCL is not touched at all, so the whole multi-context path of the host side (threads spread over contexts,
 * one merge, every context reporting the job-wide result) also runs on a one-GPU machine. */
int rccl_group_allreduce(ntsm_ctx *const *ctxs, int n)
{
	const size_t words = (size_t) ctxs[0]->n_kmers + 4;
	std::vector<int> leader(n);                            /* index of the first context on the same device */
	std::vector<ntsm_ctx *> leaders;
	for (int i = 0; i < n; ++i) {
		leader[i] = i;
		for (int j = 0; j < i; ++j)
			if (ctxs[j]->device == ctxs[i]->device) { leader[i] = leader[j]; break; }
		if (leader[i] == i) leaders.push_back(ctxs[i]);
	}
	if ((int) leaders.size() < n) {
		std::vector<uint64_t> sum(words), part(words);
		for (int l = 0; l < n; ++l) {
			if (leader[l] != l) continue;
			bool shared = false;
			for (int i = l + 1; i < n; ++i) shared |= leader[i] == l;
			if (!shared) continue;
			HIPCHK(hipSetDevice(ctxs[l]->device));
			HIPCHK(hipMemcpy(sum.data(), ctxs[l]->d_vec, words * sizeof(uint64_t), hipMemcpyDeviceToHost));
			for (int i = l + 1; i < n; ++i) {
				if (leader[i] != l) continue;
				HIPCHK(hipMemcpy(part.data(), ctxs[i]->d_vec, words * sizeof(uint64_t), hipMemcpyDeviceToHost));
				for (size_t w = 0; w < words; ++w) sum[w] += part[w];
			}
			HIPCHK(hipMemcpy(ctxs[l]->d_vec, sum.data(), words * sizeof(uint64_t), hipMemcpyHostToDevice));
		}
	}
	if (leaders.size() > 1) {
		const int rc = rccl_between_devices(leaders.data(), (int) leaders.size());
		if (rc) return rc;                                  /* nothing has been imported: every context still reports its own counts */
	}
	for (int i = 0; i < n; ++i) {
		if (leader[i] == i) continue;
		HIPCHK(hipSetDevice(ctxs[i]->device));
		HIPCHK(hipMemcpy(ctxs[i]->d_vec, ctxs[leader[i]]->d_vec, words * sizeof(uint64_t), hipMemcpyDeviceToDevice));
	}
	return NTSM_OK;
}

} // namespace ntsm_rt
