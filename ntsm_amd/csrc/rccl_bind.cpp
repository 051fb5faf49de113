/*
 * rccl_bind.cpp -- the one collective of the count path: SUM all-reduce of every context's dense count vector + totals
 * (uint64[n_kmers + 4]) across the GPUs of ONE process (ntsm_allreduce; `ntsmCount -g a,b`), RCCL over xGMI.
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

int rccl_group_allreduce(ntsm_ctx *const *ctxs, int n)
{
	const Rccl &rccl = rccl_bind();
	if (!rccl.ok) return NTSM_ERR_RCCL;
	std::vector<int> devs(n);
	for (int i = 0; i < n; ++i) devs[i] = ctxs[i]->device;
	for (int i = 0; i < n; ++i)
		for (int j = 0; j < i; ++j)
			if (devs[i] == devs[j]) return NTSM_ERR_ARG;    /* one context per device */
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

} // namespace ntsm_rt
