// transport.hpp — the collectives behind cl_exchange (include/colord_hip.h) for the multi-GPU host of the command-line compressor:
// one host THREAD per GPU in one process (the reference's orchestrator is one C++ process too, compression.cpp:547-689).
//   RcclTransport  RCCL directly (librccl: ncclCommInitAll, one communicator and one stream per rank thread): the v-collectives as ONE
//                  group of point-to-point sends / receives per call (xGMI is point to point: a grouped send / recv pattern is what its
//                  rings carry anyway), no padding to the largest shard, one stream synchronisation per call;
//   HostTransport  the same three calls through pinned host staging buffers and a barrier of the rank threads: any rank -> device map,
//                  several ranks on ONE GPU included (RCCL refuses that) — what the tests on a one-GPU box drive, and a fallback where
//                  peer access is not to be had.
// The library says WHAT is exchanged (csrc/stream.hip); these classes only move bytes.
#pragma once
#include "colord_hip.h"
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

struct Transport {
	uint32_t rank = 0, world = 1; std::string err; uint64_t bytes_moved = 0;
	virtual ~Transport() {}
	virtual cl_status all_gather_host(const uint64_t* h_vals, uint32_t n, uint64_t* h_out) = 0;
	virtual cl_status all_to_all_v(const void* d_send, const uint64_t* h_send_bytes, void* d_recv, const uint64_t* h_recv_bytes) = 0;
	virtual cl_status all_gather_v(const void* d_send, uint64_t send_bytes, void* d_recv, const uint64_t* h_recv_bytes) = 0;
	cl_status fail(const std::string& m) { err = m; return CL_E_HIP; }
	// the three callbacks of cl_exchange over `this`
	static cl_status cb_gather_host(void* u, const uint64_t* v, uint32_t n, uint64_t* o) { return ((Transport*)u)->all_gather_host(v, n, o); }
	static cl_status cb_all_to_all_v(void* u, const void* s, const uint64_t* sb, void* r, const uint64_t* rb) { return ((Transport*)u)->all_to_all_v(s, sb, r, rb); }
	static cl_status cb_all_gather_v(void* u, const void* s, uint64_t sb, void* r, const uint64_t* rb) { return ((Transport*)u)->all_gather_v(s, sb, r, rb); }
	cl_exchange exchange() { cl_exchange x; memset(&x, 0, sizeof(x)); x.user = this; x.rank = rank; x.world = world; x.all_gather_host = cb_gather_host; x.all_to_all_v = cb_all_to_all_v; x.all_gather_v = cb_all_gather_v; return x; }
};

// ---- RCCL ---------------------------------------------------------------------------------------------------------------------
// The communicators of all rank threads of the process.  A rank that fails inside a collective (a send / recv refused, a copy, a
// synchronisation) must not leave its peers waiting in theirs for a partner that will never come.  EVERY RANK ABORTS ITS OWN
// COMMUNICATOR, and only its own (round 6; round 5 let the failing thread abort all of them — a peer that was not inside a collective at
// that moment would later have handed RCCL a freed communicator): the failing rank raises the group's flag and aborts its communicator;
// a peer sees the flag at the start of its next call, or while it waits for its stream (the wait polls), aborts its own communicator —
// which ends its pending operations with an error — and returns the failure to the library.  After an abort a communicator is gone (no
// ncclCommDestroy); `comms` is only read again by destroy_all, after the rank threads have ended.
struct RcclGroup {
	std::vector<ncclComm_t> comms; std::mutex mu; std::atomic<bool> aborted{ false };
	// the owner thread of `c` gives it up (c is set to null: nobody uses it again)
	void abort_one(ncclComm_t& c)
	{
		std::lock_guard<std::mutex> l(mu);
		if (!c) return;
		for (ncclComm_t& x : comms) if (x == c) x = nullptr;
		(void)ncclCommAbort(c); c = nullptr;
	}
	void destroy_all() { std::lock_guard<std::mutex> l(mu); for (ncclComm_t& c : comms) if (c) { (void)ncclCommDestroy(c); c = nullptr; } }
};
struct RcclTransport : Transport {
	ncclComm_t comm = nullptr; hipStream_t stream = nullptr; int device = 0; RcclGroup* group = nullptr;
	uint64_t* d_small = nullptr; uint64_t small_cap = 0;                       // staging of all_gather_host
	cl_status init(ncclComm_t c, int dev, uint32_t r, uint32_t w, RcclGroup* g = nullptr)
	{
		comm = c; device = dev; rank = r; world = w; group = g;
		if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return fail("RCCL transport: no stream");
		return CL_OK;
	}
	~RcclTransport() override { if (d_small) (void)hipFree(d_small); if (stream) (void)hipStreamDestroy(stream); }
	// this rank gives up: the flag for the peers, its own communicator aborted
	void give_up() { if (group) { group->aborted.store(true); group->abort_one(comm); } }
	cl_status fail_all(const std::string& m) { const cl_status s = fail(m); give_up(); return s; }
	// before every use of `comm`: a peer has failed (or this rank has, earlier) -> no RCCL call is made any more
	bool alive(const char* what)
	{
		if (comm && !(group && group->aborted.load())) return true;
		give_up();
		fail(std::string(what) + ": the communicators were aborted by a rank that failed");
		return false;
	}
	cl_status nc(ncclResult_t r, const char* what) { if (r == ncclSuccess) return CL_OK; return fail_all(std::string(what) + ": " + ncclGetErrorString(r)); }
	// inside ncclGroupStart .. ncclGroupEnd: the failure is only noted — the group is closed first, then the rank gives up (grouped())
	cl_status ncq(ncclResult_t r, const char* what) { if (r == ncclSuccess) return CL_OK; return fail(std::string(what) + ": " + ncclGetErrorString(r)); }
	template<class F> cl_status grouped(const char* what, F&& body)
	{
		cl_status s = ncq(ncclGroupStart(), "ncclGroupStart");
		if (s != CL_OK) { give_up(); return s; }
		s = body();
		const std::string first = err;
		const cl_status e = ncq(ncclGroupEnd(), "ncclGroupEnd");                // (always closed: a group left open would swallow the next call)
		if (s != CL_OK) { err = first; give_up(); return s; }
		if (e != CL_OK) { give_up(); return e; }
		return finish(what);
	}
	// waits for the stream — polling, so that a peer's failure ends the wait; an asynchronous error of the communicator (a link went down)
	// is an error of this call
	cl_status finish(const char* what)
	{
		for (;;)
		{
			const hipError_t q = hipStreamQuery(stream);
			if (q == hipSuccess) break;
			if (q != hipErrorNotReady) { (void)hipGetLastError(); return fail_all(std::string(what) + ": synchronise"); }
			(void)hipGetLastError();
			if (group && group->aborted.load()) { give_up(); (void)hipStreamSynchronize(stream); (void)hipGetLastError(); return fail(std::string(what) + ": the communicators were aborted by a rank that failed"); }
			std::this_thread::sleep_for(std::chrono::microseconds(50));
		}
		if (!alive(what)) return CL_E_HIP;
		ncclResult_t ar = ncclSuccess;
		if (ncclCommGetAsyncError(comm, &ar) != ncclSuccess || ar != ncclSuccess) return fail_all(std::string(what) + ": " + ncclGetErrorString(ar));
		return CL_OK;
	}
	cl_status all_gather_host(const uint64_t* h_vals, uint32_t n, uint64_t* h_out) override
	{
		(void)hipSetDevice(device);
		if (!alive("all_gather_host")) return CL_E_HIP;
		const uint64_t need = (uint64_t)n * (world + 1);
		if (need > small_cap) { if (d_small) (void)hipFree(d_small); small_cap = need + 1024; if (hipMalloc((void**)&d_small, small_cap * 8) != hipSuccess) return fail_all("RCCL transport: hipMalloc"); }
		if (n && hipMemcpyAsync(d_small, h_vals, (uint64_t)n * 8, hipMemcpyHostToDevice, stream) != hipSuccess) return fail_all("RCCL transport: copy in");
		if (n) { const cl_status s = nc(ncclAllGather(d_small, d_small + n, (size_t)n * 8, ncclChar, comm, stream), "ncclAllGather"); if (s != CL_OK) return s; }
		if (n && hipMemcpyAsync(h_out, d_small + n, (uint64_t)n * 8 * world, hipMemcpyDeviceToHost, stream) != hipSuccess) return fail_all("RCCL transport: copy out");
		return finish("all_gather_host");
	}
	cl_status all_to_all_v(const void* d_send, const uint64_t* sb, void* d_recv, const uint64_t* rb) override
	{
		(void)hipSetDevice(device);
		if (!alive("all_to_all_v")) return CL_E_HIP;
		return grouped("all_to_all_v", [&]() -> cl_status {
			cl_status s = CL_OK; uint64_t so = 0, ro = 0;
			for (uint32_t p = 0; p < world && s == CL_OK; ++p)
			{	// (the share for itself travels as a send / recv pair too: one code path, RCCL makes it a local copy)
				if (sb[p]) s = ncq(ncclSend((const char*)d_send + so, (size_t)sb[p], ncclChar, (int)p, comm, stream), "ncclSend");
				if (s == CL_OK && rb[p]) s = ncq(ncclRecv((char*)d_recv + ro, (size_t)rb[p], ncclChar, (int)p, comm, stream), "ncclRecv");
				so += sb[p]; ro += rb[p]; bytes_moved += p == rank ? 0 : sb[p];
			}
			return s;
		});
	}
	cl_status all_gather_v(const void* d_send, uint64_t send_bytes, void* d_recv, const uint64_t* rb) override
	{
		(void)hipSetDevice(device);
		if (!alive("all_gather_v")) return CL_E_HIP;
		if (rb[rank] != send_bytes) return fail_all("all_gather_v: send_bytes != h_recv_bytes[rank]");
		return grouped("all_gather_v", [&]() -> cl_status {
			cl_status s = CL_OK; uint64_t ro = 0;
			for (uint32_t p = 0; p < world && s == CL_OK; ++p)
			{
				if (send_bytes) s = ncq(ncclSend(d_send, (size_t)send_bytes, ncclChar, (int)p, comm, stream), "ncclSend");
				if (s == CL_OK && rb[p]) s = ncq(ncclRecv((char*)d_recv + ro, (size_t)rb[p], ncclChar, (int)p, comm, stream), "ncclRecv");
				ro += rb[p]; bytes_moved += p == rank ? 0 : send_bytes;
			}
			return s;
		});
	}
};

// ---- host staging ----------------------------------------------------------------------------------------------------------------
// Every call passes BOTH of its barriers whatever happens to it (a rank that returned early would leave the others in the second one
// for ever); a failure is recorded in the hub, the peers see it after the first barrier, skip their copies and fail too.
struct HostHub {
	uint32_t world; std::mutex mu; std::condition_variable cv; uint32_t arrived = 0; uint64_t epoch = 0;
	std::atomic<bool> failed{ false };
	std::vector<std::vector<uint64_t>> small;                                  // per rank: what it contributes to all_gather_host
	std::vector<uint8_t*> stage; std::vector<uint64_t> stage_cap; std::vector<std::vector<uint64_t>> counts;   // per rank: pinned copy of its send buffer, bytes per destination
	explicit HostHub(uint32_t w) : world(w), small(w), stage(w, nullptr), stage_cap(w, 0), counts(w) {}
	~HostHub() { for (uint8_t* p : stage) if (p) (void)hipHostFree(p); }
	void barrier()
	{
		std::unique_lock<std::mutex> l(mu);
		const uint64_t e = epoch;
		if (++arrived == world) { arrived = 0; ++epoch; cv.notify_all(); }
		else cv.wait(l, [&] { return epoch != e; });
	}
};
struct HostTransport : Transport {
	HostHub* hub = nullptr; int device = 0;
	void init(HostHub* h, int dev, uint32_t r) { hub = h; device = dev; rank = r; world = h->world; }
	cl_status fail_all(const std::string& m) { hub->failed.store(true); return fail(m); }
	cl_status peer_failed(const char* what) { return fail(std::string(what) + ": another rank failed"); }
	cl_status publish(const void* d_send, uint64_t bytes)
	{
		(void)hipSetDevice(device);
		if (bytes > hub->stage_cap[rank])
		{
			if (hub->stage[rank]) (void)hipHostFree(hub->stage[rank]);
			hub->stage[rank] = nullptr; hub->stage_cap[rank] = 0;
			void* p = nullptr;
			if (hipHostMalloc(&p, bytes + bytes / 4 + 4096, hipHostMallocPortable) != hipSuccess) return fail_all("host transport: hipHostMalloc");
			hub->stage[rank] = (uint8_t*)p; hub->stage_cap[rank] = bytes + bytes / 4 + 4096;
		}
		if (bytes && hipMemcpy(hub->stage[rank], d_send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail_all("host transport: copy to host");
		return CL_OK;
	}
	cl_status all_gather_host(const uint64_t* h_vals, uint32_t n, uint64_t* h_out) override
	{
		hub->small[rank].assign(h_vals, h_vals + n);
		hub->barrier();
		cl_status s = CL_OK;
		for (uint32_t p = 0; p < world && s == CL_OK; ++p)
		{
			if (hub->small[p].size() != n) s = fail_all("all_gather_host: ranks disagree on n");
			else memcpy(h_out + (uint64_t)p * n, hub->small[p].data(), (uint64_t)n * 8);
		}
		hub->barrier();                                                         // (nobody overwrites its contribution before everybody has read it)
		if (s == CL_OK && hub->failed.load()) s = peer_failed("all_gather_host");
		return s;
	}
	cl_status all_to_all_v(const void* d_send, const uint64_t* sb, void* d_recv, const uint64_t* rb) override
	{
		uint64_t tot = 0; for (uint32_t p = 0; p < world; ++p) tot += sb[p];
		cl_status s = publish(d_send, tot);
		hub->counts[rank].assign(sb, sb + world);
		hub->barrier();
		(void)hipSetDevice(device);
		if (s == CL_OK && hub->failed.load()) s = peer_failed("all_to_all_v");  // (a rank whose publish failed has no staging buffer to read from)
		uint64_t ro = 0;
		for (uint32_t p = 0; p < world && s == CL_OK; ++p)
		{	// what rank p holds for this rank: behind its shares for the ranks before this one
			uint64_t so = 0; for (uint32_t q = 0; q < rank; ++q) so += hub->counts[p][q];
			if (hub->counts[p][rank] != rb[p]) { s = fail_all("all_to_all_v: ranks disagree on a share's size"); break; }
			if (rb[p] && hipMemcpy((char*)d_recv + ro, hub->stage[p] + so, rb[p], hipMemcpyHostToDevice) != hipSuccess) s = fail_all("host transport: copy to device");
			ro += rb[p]; bytes_moved += p == rank ? 0 : rb[p];
		}
		hub->barrier();
		if (s == CL_OK && hub->failed.load()) s = peer_failed("all_to_all_v");
		return s;
	}
	cl_status all_gather_v(const void* d_send, uint64_t send_bytes, void* d_recv, const uint64_t* rb) override
	{
		cl_status s = rb[rank] != send_bytes ? fail_all("all_gather_v: send_bytes != h_recv_bytes[rank]") : publish(d_send, send_bytes);
		hub->barrier();
		(void)hipSetDevice(device);
		if (s == CL_OK && hub->failed.load()) s = peer_failed("all_gather_v");
		uint64_t ro = 0;
		for (uint32_t p = 0; p < world && s == CL_OK; ++p)
		{
			if (rb[p] && hipMemcpy((char*)d_recv + ro, hub->stage[p], rb[p], hipMemcpyHostToDevice) != hipSuccess) s = fail_all("host transport: copy to device");
			ro += rb[p]; bytes_moved += p == rank ? 0 : rb[p];
		}
		hub->barrier();
		if (s == CL_OK && hub->failed.load()) s = peer_failed("all_gather_v");
		return s;
	}
};
