// common.hpp — context, device memory helpers and wave64 device primitives shared by the kernels.
// gfx950 only: wavefront = 64 lanes, no dual paths.
#pragma once
// device functions that a debugging build (tests/tools, -DCL_HOST_DEBUG) also compiles for the host; the library never does
#ifdef CL_HOST_DEBUG
#define CL_DEV __host__ __device__
#else
#define CL_DEV __device__
#endif
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <map>
#include <deque>
#include <mutex>
#include <atomic>
#include <condition_variable>
#include <thread>
#include <chrono>
#include "../../include/colord_hip.h"

#define CL_WAVE 64

struct KernelTime { double ms = 0; uint32_t launches = 0; double bytes = 0, cells = 0; };   // bytes = algorithmic HBM bytes of the timed launches; cells = DP cell updates (aligners: rows x columns of their gaps)

// Device memory of a context: a sub-allocator over a few large slabs.  hipMalloc / hipFree cost 0.1-1 ms each (a second
// for tens of GB) and synchronise the device, and a cache of whole hipMalloc blocks keyed by size wastes HBM exactly where
// this pipeline needs it (stages ask for 34, 19, 16, 15, 10, 9 ... GB one after the other; a block kept for one size serves
// the next badly).  So: blocks are carved best-fit, at their exact size, out of slabs obtained from the driver; free extents
// coalesce; a request no extent holds adds a slab (its own size if it is large, else a quarter of what the pool holds already,
// 256 MB to 4 GB).  Several contexts work on one GPU at a time (the quality stream's, the encode lanes and the preparation threads of
// cl_compressor): they SHARE the pool of their device (cl_device_pool).  With a pool per context — seven of them by round 3 — the
// slabs each one kept for its own peak added up to more than the device holds, and every chunk paid hipFree / hipMalloc pairs of
// several GB (a device-wide synchronisation each: the encode lanes stood still for half of their time).  One pool holds the peak of
// the SUM of what is live.  When the device runs short all the same, slabs that are entirely free are given back.
struct cl_ctx;
void cl_ctx_drain(cl_ctx* c);                  // capi.hip: waits for every stream of the context
void cl_ctx_set_priority(cl_ctx* c, int level, int role = -1); // capi.hip: +1 highest, -1 lowest, 0 default stream priority of the context; role: CL_ROLE_* (-1: unchanged)
int cl_ctx_fence(cl_ctx* c, hipEvent_t* ev);   // capi.hip: records an event on every stream of the context; returns their number (<= 4)
// COLORD_HIP_POOL_POISON=1 (debugging the hand-over of memory between contexts): every block is filled with a pattern on the releasing
// context's main stream when it goes back to the pool, and checked when it is carved for ANOTHER context (after the owner's fence says
// so): a broken pattern = somebody wrote into the block after its release.  The other direction shows by itself: whoever still READS a
// block it released (a side stream that was not joined) now reads the pattern, and the byte comparisons of the suite fail.
void cl_ctx_poison(cl_ctx* c, void* p, uint64_t bytes);       // capi.hip: hipMemsetAsync on the context's main stream
uint64_t cl_pool_poison_check(void* p, uint64_t bytes);      // capi.hip: waits for the device, returns the number of 4-byte words that lost the pattern
static inline bool cl_pool_poison() { static const bool on = getenv("COLORD_HIP_POOL_POISON") != nullptr; return on; }
struct DevPool {
	// A free extent remembers who released it and when (pool clock).  Its owner may have it back at once — a context's own reuse
	// is ordered by its streams, as with a pool per context; anybody else only once the owner's streams have drained since
	// (`drained`): kernels of the owner may still be reading a block it released early.
	struct Ext { uint64_t len; int32_t owner; uint64_t clock; bool poisoned = false; };   // poisoned: the whole extent holds the pattern (COLORD_HIP_POOL_POISON)
	struct Slab { char* base = nullptr; uint64_t size = 0, free_bytes = 0; std::map<uint64_t, Ext> ext; };   // ext: offset -> free extent
	std::mutex mu;                                // a buffer made on one thread (an encode lane of cl_compressor) may be released on another
	std::vector<Slab> slabs;
	std::vector<cl_ctx*> owners; std::vector<uint64_t> drained; uint64_t clock = 0;      // by owner id
	std::vector<uint32_t> pinned; std::condition_variable pin_cv;                          // drains of an owner's streams in progress outside the mutex (drop_owner waits for them)
	// fences: every few releases of an owner, an event on each of its streams.  What it released before a fence is anybody's once
	// the fence's events have completed — no need to wait for whatever the owner has started since.
	struct Fence { uint64_t clock; hipEvent_t ev[4]; int n; };
	std::vector<std::deque<Fence>> fences; std::vector<uint32_t> since_fence; std::vector<hipEvent_t> spare_events;
	uint64_t reserved = 0, live_bytes = 0, peak_live = 0, peak_total = 0; uint32_t n_mallocs = 0, n_drains = 0;   // (statistics for COLORD_HIP_POOL_DEBUG)
	uint64_t n_poison_checks = 0, n_poison_bad = 0;                                                                  // (COLORD_HIP_POOL_POISON)
	static constexpr uint64_t ALIGN = 256, PAD = 256, SLAB_MIN = 256ull << 20, SLAB_MAX = 4ull << 30;
	int32_t add_owner(cl_ctx* c) { std::lock_guard<std::mutex> l(mu); owners.push_back(c); drained.push_back(0); fences.emplace_back(); since_fence.push_back(0); pinned.push_back(0); return (int32_t)owners.size() - 1; }
	void drop_owner(int32_t id)
	{	// (the caller has drained its streams: nothing of it is in flight; they are destroyed after this returns)
		std::unique_lock<std::mutex> l(mu);
		if (id < 0 || (size_t)id >= owners.size()) return;
		pin_cv.wait(l, [&] { return pinned[id] == 0; });
		owners[id] = nullptr; drained[id] = ~0ull;
		for (auto& f : fences[id]) for (int i = 0; i < f.n; ++i) spare_events.push_back(f.ev[i]);
		fences[id].clear();
	}
	// the fences of `id` whose events have completed move `drained` forward
	void poll_fences_locked(int32_t id)
	{
		auto& q = fences[id];
		while (!q.empty())
		{
			bool done = true;
			for (int i = 0; i < q.front().n && done; ++i) done = hipEventQuery(q.front().ev[i]) == hipSuccess;
			if (!done) { (void)hipGetLastError(); break; }
			if (drained[id] < q.front().clock) drained[id] = q.front().clock;
			for (int i = 0; i < q.front().n; ++i) spare_events.push_back(q.front().ev[i]);
			q.pop_front();
		}
	}
	bool clean(const Ext& e) const { return e.owner < 0 || e.clock <= drained[e.owner]; }
	bool usable(const Ext& e, int32_t who) const { return e.owner == who || clean(e); }
	// gives entirely free slabs back to the driver, largest first, until `want` bytes are freed; returns the bytes freed
	uint64_t shed_locked(uint64_t want)
	{
		uint64_t freed = 0;
		while (freed < want)
		{
			size_t best = slabs.size();
			for (size_t i = 0; i < slabs.size(); ++i) if (slabs[i].free_bytes == slabs[i].size && (best == slabs.size() || slabs[i].size > slabs[best].size)) best = i;
			if (best == slabs.size()) break;
			(void)hipFree(slabs[best].base); freed += slabs[best].size; reserved -= slabs[best].size;     // (hipFree waits for the device: whatever was in flight on the slab is through)
			slabs.erase(slabs.begin() + best);
		}
		return freed;
	}
	// makes room for a new slab of r bytes when the device is short of it
	void make_room(uint64_t r, bool dbg)
	{
		size_t fr = 0, tot = 0;
		if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return; }
		const uint64_t need = r + (1ull << 30);
		if (fr >= need) return;
		const uint64_t freed = shed_locked(need - fr);                          // (pools of other devices hold nothing this device could use)
		if (dbg) fprintf(stderr, "[pool] device has %.1f GB free, %.3f GB wanted: gave back %.3f GB of free slabs\n", fr / 1e9, r / 1e9, freed / 1e9);
	}
	// neighbours that have become compatible since they were released (their owners drained) are joined
	void coalesce_locked()
	{
		for (Slab& S : slabs)
			for (auto it = S.ext.begin(); it != S.ext.end();)
			{
				auto nx = std::next(it);
				if (nx == S.ext.end()) break;
				if (it->first + it->second.len == nx->first && (it->second.owner == nx->second.owner || (clean(it->second) && clean(nx->second))))
				{
					if (it->second.owner != nx->second.owner) { it->second.owner = -1; it->second.clock = 0; } else it->second.clock = std::max(it->second.clock, nx->second.clock);
					it->second.poisoned = it->second.poisoned && nx->second.poisoned;
					it->second.len += nx->second.len; S.ext.erase(nx);
				}
				else it = nx;
			}
	}
	// waits for the streams of owner `id` with the mutex released; the owner is pinned meanwhile (drop_owner, hence the destruction
	// of its context, waits for the pin)
	void drain_unlocked(std::unique_lock<std::mutex>& lock, int32_t id)
	{
		cl_ctx* oc = owners[id];
		if (!oc) return;
		++pinned[id];
		lock.unlock();
		cl_ctx_drain(oc);
		lock.lock();
		if (--pinned[id] == 0) pin_cv.notify_all();
	}
	// best fit among the extents `who` may use; *foreign: the owner of a fitting extent it may not use yet (or -1)
	bool carve(uint64_t r, int32_t who, void** out, int32_t* foreign)
	{
		size_t bs = slabs.size(); uint64_t boff = 0, blen = ~0ull;
		if (foreign) *foreign = -1;
		for (size_t i = 0; i < slabs.size(); ++i)
		{
			if (slabs[i].free_bytes < r) continue;
			for (auto& e : slabs[i].ext)
			{
				if (e.second.len < r) continue;
				if (!usable(e.second, who)) { if (foreign && *foreign < 0) *foreign = e.second.owner; continue; }
				if (e.second.len < blen) { bs = i; boff = e.first; blen = e.second.len; if (blen == r) break; }
			}
			if (blen == r) break;
		}
		if (bs == slabs.size()) return false;
		Slab& S = slabs[bs];
		const Ext old = S.ext[boff];
		S.ext.erase(boff);
		if (blen > r) S.ext.emplace(boff + r, Ext{ blen - r, old.owner, old.clock, old.poisoned });
		S.free_bytes -= r;
		*out = S.base + boff;
		if (old.poisoned && old.owner != who && cl_pool_poison())
		{	// handed to another context: the pattern must have survived since the release
			const uint64_t bad = cl_pool_poison_check(*out, r);
			++n_poison_checks;
			if (bad) { ++n_poison_bad; fprintf(stderr, "colord_hip: POOL POISON: a block of %llu bytes released by context %d (clock %llu) was written to after its release: %llu words differ; now handed to context %d\n", (unsigned long long)r, old.owner, (unsigned long long)old.clock, (unsigned long long)bad, who); }
		}
		return true;
	}
	hipError_t get(uint64_t bytes, void** out, uint64_t* got, int32_t who)
	{
		static const bool dbg = getenv("COLORD_HIP_POOL_DEBUG") != nullptr;
		std::unique_lock<std::mutex> lock(mu);
		const uint64_t r = (bytes + ALIGN - 1) / ALIGN * ALIGN + PAD;          // (the pad keeps a kernel's vector load past its last element inside the block)
		*got = r;
		int32_t foreign = -1;
		bool ok = carve(r, who, out, &foreign);
		if (!ok)
		{
			for (size_t i = 0; i < owners.size(); ++i) if (owners[i]) poll_fences_locked((int32_t)i);
			coalesce_locked();
			ok = carve(r, who, out, &foreign);
		}
		for (int tries = 0; !ok && foreign >= 0 && tries < 8; ++tries)
		{	// memory that would do was released by another context whose kernels may still be using it: wait for that context's streams
			// (not the device), then it is anybody's
			const uint64_t upto = clock;
			++n_drains;
			drain_unlocked(lock, foreign);
			if (drained[foreign] < upto) drained[foreign] = upto;
			coalesce_locked();
			ok = carve(r, who, out, &foreign);
		}
		if (!ok)
		{
			const uint64_t g = 2ull << 20;
			// Few, large slabs: a new one is as large as everything the pool holds already (1 GB at least, 48 GB at most, a
			// large request with an eighth to spare: request sizes creep from chunk to chunk), within what the device has free.
			uint64_t sz = std::max<uint64_t>(r + r / 8, std::min<uint64_t>(std::max<uint64_t>(reserved, 1ull << 30), 48ull << 30));
			{
				size_t fr = 0, tot = 0;
				if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr > (3ull << 30) && sz > fr - (2ull << 30)) sz = std::max<uint64_t>(r, fr - (2ull << 30));
				else (void)hipGetLastError();
			}
			sz = (sz + g - 1) / g * g;
			make_room(sz, dbg);
			void* base = nullptr;
			const auto t_slab = std::chrono::steady_clock::now();
			hipError_t e = hipMalloc(&base, sz);
			if (e != hipSuccess && sz > r + g)
			{	// not even after making room: the request alone
				(void)hipGetLastError();
				sz = (r + g - 1) / g * g;
				make_room(sz, dbg);
				e = hipMalloc(&base, sz);
			}
			++n_mallocs;
			if (dbg) fprintf(stderr, "[pool] slab of %.3f GB (%s, %.1f ms) for a block of %.3f GB; live %.3f GB, reserved %.3f GB in %zu slabs\n", sz / 1e9, e == hipSuccess ? "ok" : "failed",
				std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_slab).count(), r / 1e9, live_bytes / 1e9, reserved / 1e9, slabs.size());
			if (e != hipSuccess)
			{	// The device is full.  Other contexts of this process hold most of it for a moment only (the look-ahead stages of
				// cl_compressor): wait for what they release — up to a few seconds — before this becomes the caller's error.
				(void)hipGetLastError();
				bool got_it = false;
				for (int tries = 0; tries < 150 && !got_it; ++tries)
				{
					lock.unlock();
					std::this_thread::sleep_for(std::chrono::milliseconds(20));
					lock.lock();
					for (size_t i = 0; i < owners.size(); ++i) if (owners[i]) poll_fences_locked((int32_t)i);
					coalesce_locked();
					int32_t fo = -1;
					got_it = carve(r, who, out, &fo);
					if (!got_it && fo >= 0 && (tries & 7) == 7)
					{	// (something fits but its owner has not passed a fence since: wait for that owner's streams)
						const uint64_t upto = clock;
						drain_unlocked(lock, fo);
						if (drained[fo] < upto) drained[fo] = upto;
					}
				}
				if (!got_it) return e;
				if (dbg) fprintf(stderr, "[pool] no slab of %.3f GB to be had: waited for a block of %.3f GB\n", sz / 1e9, r / 1e9);
				live_bytes += r; if (live_bytes > peak_live) peak_live = live_bytes;
				return hipSuccess;
			}
			Slab S; S.base = (char*)base; S.size = sz; S.free_bytes = sz; S.ext.emplace(0, Ext{ sz, -1, 0 });
			slabs.push_back(std::move(S));
			reserved += sz; if (reserved > peak_total) peak_total = reserved;
			if (!carve(r, who, out, nullptr)) return hipErrorOutOfMemory;
		}
		live_bytes += r; if (live_bytes > peak_live) peak_live = live_bytes;
		return hipSuccess;
	}
	void put(void* p, uint64_t r, int32_t who)
	{
		std::lock_guard<std::mutex> lock(mu);
		for (Slab& S : slabs)
		{
			if ((char*)p < S.base || (char*)p >= S.base + S.size) continue;
			uint64_t off = (uint64_t)((char*)p - S.base);
			Ext me{ r, who, ++clock };
			if (cl_pool_poison() && who >= 0 && owners[who]) { cl_ctx_poison(owners[who], p, r); me.poisoned = true; }
			auto nx = S.ext.lower_bound(off);
			if (nx != S.ext.begin()) { auto pv = std::prev(nx); if (pv->first + pv->second.len == off && pv->second.owner == who) { off = pv->first; me.len += pv->second.len; me.poisoned = me.poisoned && pv->second.poisoned; S.ext.erase(pv); } }
			if (nx != S.ext.end() && off + me.len == nx->first && nx->second.owner == who) { me.len += nx->second.len; me.poisoned = me.poisoned && nx->second.poisoned; S.ext.erase(nx); }
			S.ext.emplace(off, me);
			S.free_bytes += r; live_bytes -= r;
			if (who >= 0 && owners[who] && (++since_fence[who] >= 8 || r >= (64ull << 20)))
			{
				since_fence[who] = 0;
				poll_fences_locked(who);
				if (fences[who].size() < 64)
				{
					Fence f; f.clock = clock; f.n = 0;
					int have = 0;                                                      // events actually obtained (a fence needs all four)
					for (; have < 4; ++have) { if (spare_events.empty()) { hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); break; } spare_events.push_back(e); } f.ev[have] = spare_events.back(); spare_events.pop_back(); }
					if (have == 4) f.n = cl_ctx_fence(owners[who], f.ev);
					for (int i = f.n; i < have; ++i) spare_events.push_back(f.ev[i]);
					if (f.n > 0) fences[who].push_back(f);
				}
			}
			return;
		}
		if (reserved) fprintf(stderr, "colord_hip: pool released a block it does not own\n");
	}
	void trim()
	{
		std::lock_guard<std::mutex> lock(mu);
		if (cl_pool_poison()) fprintf(stderr, "[pool poison] %llu blocks checked at their hand-over to another context, %llu had lost the pattern\n", (unsigned long long)n_poison_checks, (unsigned long long)n_poison_bad);
		if (getenv("COLORD_HIP_POOL_DEBUG")) fprintf(stderr, "[pool] at trim: peak live %.1f GB, peak reserved %.1f GB, reserved %.1f GB in %zu slabs, %u hipMalloc calls, %u waits for another context's streams, still live %.3f GB\n", peak_live / 1e9, peak_total / 1e9, reserved / 1e9, slabs.size(), n_mallocs, n_drains, live_bytes / 1e9);
		for (auto& S : slabs) (void)hipFree(S.base);
		slabs.clear(); reserved = 0;
	}
};

static inline bool cl_pool_debug() { static const bool dbg = getenv("COLORD_HIP_POOL_DEBUG") != nullptr; return dbg; }

// the pool of a device, shared by every context on it; given back to the driver when the last of them goes
struct DevicePools { std::mutex mu; std::map<int, std::pair<DevPool*, int>> by_dev; };
inline DevicePools& cl_device_pools() { static DevicePools* p = new DevicePools; return *p; }
inline DevPool& cl_device_pool_acquire(int device)
{
	DevicePools& D = cl_device_pools();
	std::lock_guard<std::mutex> l(D.mu);
	auto& e = D.by_dev[device];
	if (!e.first) e.first = new DevPool();
	++e.second;
	return *e.first;
}
inline void cl_device_pool_release(int device)
{
	DevicePools& D = cl_device_pools();
	std::lock_guard<std::mutex> l(D.mu);
	auto it = D.by_dev.find(device);
	if (it == D.by_dev.end()) return;
	if (--it->second.second == 0) it->second.first->trim();                   // (the pool object stays: buffers released late still find it)
}

struct cl_ctx {
	DevPool& pool;
	int device = 0;
	int32_t pool_id = -1;                        // this context as an owner of free extents of the shared pool
	explicit cl_ctx(int dev) : pool(cl_device_pool_acquire(dev)), device(dev) { pool_id = pool.add_owner(this); }
	hipStream_t stream = nullptr;                // the context's main stream; never reassigned while the context works (the pool's fences and drains read it from other threads)
	hipStream_t launch = nullptr;                // owner thread only: where LAUNCH and its timing events go while a stage works on a side / coder stream (null: `stream`)
	int prio = 0;                                // priority of this context's streams (cl_ctx_set_priority; 0 = the runtime's default)
	bool masked = false;                         // its streams were made on the role's CUs
	int role = 0;                                // what the context does in the compressor (CL_ROLE_*): selects its CU mask, if any (COLORD_HIP_CU_MASK)
	// further streams, created by the owner thread at first use (a stream that exists takes its turn on the runtime's hardware queues
	// whether it is used or not: created eagerly for every context they cost the 50-Gbase pass 25 %) and read by the shared pool's
	// fences and drains from other threads: atomic pointers, set once
	std::atomic<hipStream_t> side{ nullptr };     // second stream for chains that would leave the machine idle
	std::atomic<hipStream_t> side2{ nullptr };    // third stream: the four-per-wave aligner next to the tail-bound wave-per-gap one
	std::atomic<hipStream_t> side3{ nullptr };    // fourth stream: the giant gaps
	std::string err;
	bool timing = false;
	std::map<std::string, KernelTime> times;     // per-kernel accumulated HIP-event time of the last API call
	std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> pending;
	std::vector<double> pending_bytes, pending_cells;
	double next_bytes = 0, next_cells = 0;       // algorithmic bytes (LAUNCHB) / DP cells (aligners) of the next LAUNCH
	std::vector<hipEvent_t> ev_pool;
	int n_cu = 256;
	uint64_t* inv_tab = nullptr;                 // floor((2^64-1) / t) for t < 2^21: the interval coder's division table (rc_dev.hpp), made at first use
	uint64_t* slots_h = nullptr; uint64_t* slots_d = nullptr; uint32_t slot_next = 0;   // pinned, device-mapped words that kernels write results the host waits for into (cl_slot)
	struct ScanCtl { unsigned long long* p = nullptr; uint64_t words = 0, got = 0; uint32_t gen = 0; unsigned long long tickets = 0; };   // gen: generation of the last scan (0: buffer not zeroed yet); tickets: tile tickets drawn so far
	std::map<void*, ScanCtl> scan_ctl;           // status words of the look-back scans, one buffer per stream of the context (scan.hip; owner thread only)
	std::vector<cl_ctx*> lanes;                  // encode lanes of cl_compressor (contexts of their own; kept for the next compressor, freed with this context)
	cl_ctx* prep = nullptr;                      // context of cl_compressor's DNA preparation thread (same life cycle)
	cl_ctx* qprep = nullptr;                     // ... and of its quality preparation thread
};

static inline cl_status cl_fail(cl_ctx* c, cl_status s, const std::string& msg)
{
	if (c) c->err = msg;
	return s;
}

#define HIP_TRY(ctx, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
	return cl_fail((ctx), CL_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e) + " @" + __FILE__ + ":" + std::to_string(__LINE__)); } } while (0)
#define CL_TRY(expr) do { cl_status _s = (expr); if (_s != CL_OK) return _s; } while (0)

// ---- device buffers (RAII, freed with the owning object) -----------------------------------------
template<typename T> struct DevBuf {
	T* p = nullptr; uint64_t n = 0; uint64_t bytes = 0; DevPool* pool = nullptr; int32_t owner = -1;
	DevBuf() = default;
	DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
	DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n), bytes(o.bytes), pool(o.pool), owner(o.owner) { o.p = nullptr; o.n = 0; }
	DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; bytes = o.bytes; pool = o.pool; owner = o.owner; o.p = nullptr; o.n = 0; } return *this; }
	~DevBuf() { release(); }
	void release() { if (p) { if (pool) pool->put(p, bytes, owner); else (void)hipFree(p); } p = nullptr; n = 0; }
	hipError_t alloc(cl_ctx* c, uint64_t count);
};
template<typename T> hipError_t DevBuf<T>::alloc(cl_ctx* c, uint64_t count)
{
	release(); n = count; if (!count) count = 1;
	pool = &c->pool; owner = c->pool_id;
	void* q = nullptr;
	hipError_t e = pool->get(count * sizeof(T), &q, &bytes, owner);
	p = (T*)q;
	if (e != hipSuccess) { p = nullptr; n = 0; }
	return e;
}

#define DEV_ALLOC(ctx, buf, count) do { hipError_t _e = (buf).alloc((ctx), count); \
	if ((buf).bytes >= (1ull << 30) && cl_pool_debug()) fprintf(stderr, "[pool] %s:%d %s %.2f GB (live %.1f GB)\n", &__FILE__[sizeof(__FILE__) > 24 ? sizeof(__FILE__) - 24 : 0], __LINE__, #buf, (buf).bytes / 1e9, (ctx)->pool.live_bytes / 1e9); \
	if (_e != hipSuccess) \
	return cl_fail((ctx), CL_E_NOMEM, std::string("hipMalloc(" #buf ") of ") + std::to_string((uint64_t)(count)) + " elems: " + hipGetErrorString(_e)); } while (0)

// Small results the host waits for (a scan's total, a level's class bounds, a counter): kernels write them straight into pinned host
// memory that is mapped into the device's address space, and the host reads them after synchronising the stream — no copy command
// (a D2H copy into pageable memory is a dispatch of its own plus a staged transfer: ~1 ms each with six threads in the runtime, 1 700
// of them per encode lane and pass in round 3).  Slots go round a ring of 4096 words; a user synchronises before it reads, long
// before the ring comes round.  Owner thread only.
static inline hipError_t cl_slot(cl_ctx* c, uint32_t n_words, uint64_t** h, uint64_t** d)
{
	constexpr uint32_t RING = 4096;
	if (!c->slots_h)
	{
		void* p = nullptr; void* dp = nullptr;
		hipError_t e = hipHostMalloc(&p, RING * 8, hipHostMallocMapped);
		if (e != hipSuccess) return e;
		e = hipHostGetDevicePointer(&dp, p, 0);
		if (e != hipSuccess) { (void)hipHostFree(p); return e; }
		memset(p, 0, RING * 8);
		c->slots_h = (uint64_t*)p; c->slots_d = (uint64_t*)dp;
	}
	if (n_words > RING) return hipErrorInvalidValue;
	if (c->slot_next + n_words > RING) c->slot_next = 0;
	*h = c->slots_h + c->slot_next; *d = c->slots_d + c->slot_next;
	c->slot_next += n_words;
	return hipSuccess;
}

// ---- per-kernel timing with HIP events on the stream of the launch --------------------------------
static inline hipStream_t cl_launch_stream(const cl_ctx* c) { return c->launch ? c->launch : c->stream; }
// the launches of a scope on another stream of the context (owner thread only; cl_ctx::stream itself is never swapped: the shared
// pool's fences and drains read it from other threads)
struct LaunchOn { cl_ctx* c; hipStream_t prev; LaunchOn(cl_ctx* c_, hipStream_t s) : c(c_), prev(c_->launch) { c->launch = s; } ~LaunchOn() { c->launch = prev; } };
static inline bool cl_sync_debug() { static const bool d = getenv("COLORD_HIP_SYNC_DEBUG") != nullptr; return d; }
struct KernelTimer {
	cl_ctx* c; const char* name; hipEvent_t a = nullptr, b = nullptr;
	KernelTimer(cl_ctx* c_, const char* n) : c(c_), name(n)
	{
		if (!c->timing) return;
		auto get = [&]() { hipEvent_t e; if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); } else (void)hipEventCreate(&e); return e; };
		a = get(); b = get();
		(void)hipEventRecord(a, cl_launch_stream(c));
	}
	~KernelTimer()
	{
		if (cl_sync_debug())
		{	// COLORD_HIP_SYNC_DEBUG: wait for every launch and name the kernel that was running when the device reported an error
			fprintf(stderr, "[sync] > %s\n", name);
			const hipError_t e = hipStreamSynchronize(cl_launch_stream(c));
			fprintf(stderr, "[sync] < %s%s%s\n", name, e == hipSuccess ? "" : ": ", e == hipSuccess ? "" : hipGetErrorString(e));
		}
		if (!c->timing) return;
		(void)hipEventRecord(b, cl_launch_stream(c));
		c->pending.push_back({ name, { a, b } });
		c->pending_bytes.push_back(c->next_bytes); c->next_bytes = 0;
		c->pending_cells.push_back(c->next_cells); c->next_cells = 0;
	}
};
// every kernel launch goes through LAUNCH so that per-kernel HIP-event times are complete
#define LAUNCH(ctx, kernel, grid, block, ...) do { KernelTimer _kt((ctx), #kernel); \
	hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, cl_launch_stream(ctx), __VA_ARGS__); } while (0)
// a template kernel under the name of its instantiation (as rocprofv3 lists it), so that the two sets of times can be laid side by side
#define LAUNCHB_NAMED(ctx, name, bytes, kernel, grid, block, ...) do { (ctx)->next_bytes = (double)(bytes); KernelTimer _kt((ctx), (name)); \
	hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, cl_launch_stream(ctx), __VA_ARGS__); } while (0)
#define LAUNCH_NAMED(ctx, name, kernel, grid, block, ...) LAUNCHB_NAMED(ctx, name, 0.0, kernel, grid, block, __VA_ARGS__)
// LAUNCH with the algorithmic HBM byte count of this launch (for achieved-GB/s reporting)
#define LAUNCHB(ctx, bytes, kernel, grid, block, ...) do { (ctx)->next_bytes = (double)(bytes); LAUNCH(ctx, kernel, grid, block, __VA_ARGS__); } while (0)
// the same with dynamic LDS
#define LAUNCHB_SHM(ctx, bytes, kernel, grid, block, shm, ...) do { (ctx)->next_bytes = (double)(bytes); KernelTimer _kt((ctx), #kernel); \
	hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (shm), cl_launch_stream(ctx), __VA_ARGS__); } while (0)
// CU partitioning (measured in DESIGN.md 5b; off unless COLORD_HIP_CU_MASK is set): streams of a role run on a range of the device's
// CUs only, e.g. COLORD_HIP_CU_MASK="coder:0-31,lane:32-255,main:32-255,qual:32-255,prep:32-255" keeps the interval coders'
// dependent chains (k_range_code on the coders' own streams) on 32 CUs nobody else may fill.  Bits are the runtime's CU numbering
// (hipExtStreamCreateWithCUMask); a masked stream has the default priority (the API takes no priority).
enum { CL_ROLE_MAIN = 0, CL_ROLE_QUAL = 1, CL_ROLE_LANE = 2, CL_ROLE_PREP = 3, CL_ROLE_CODER = 4, CL_N_ROLES = 5 };
struct CuMaskCfg { bool any = false; bool on[CL_N_ROLES] = { false, false, false, false, false }; uint32_t lo[CL_N_ROLES] = { 0, 0, 0, 0, 0 }, hi[CL_N_ROLES] = { 0, 0, 0, 0, 0 }; };
inline const CuMaskCfg& cl_cu_mask_cfg()
{
	static const CuMaskCfg cfg = []() {
		CuMaskCfg c;
		const char* e = getenv("COLORD_HIP_CU_MASK");
		if (!e) return c;
		static const char* names[CL_N_ROLES] = { "main", "qual", "lane", "prep", "coder" };
		std::string s(e); size_t p = 0;
		while (p < s.size())
		{
			size_t q = s.find(',', p); if (q == std::string::npos) q = s.size();
			const std::string item = s.substr(p, q - p); p = q + 1;
			const size_t colon = item.find(':'), dash = item.find('-');
			if (colon == std::string::npos || dash == std::string::npos || dash < colon) continue;
			for (int r = 0; r < CL_N_ROLES; ++r) if (item.substr(0, colon) == names[r])
			{
				c.lo[r] = (uint32_t)atoi(item.substr(colon + 1, dash - colon - 1).c_str()); c.hi[r] = (uint32_t)atoi(item.substr(dash + 1).c_str());
				if (c.hi[r] >= c.lo[r] && c.hi[r] < 1024) { c.on[r] = true; c.any = true; }
			}
		}
		return c;
	}();
	return cfg;
}
// COLORD_HIP_ROLE_PRIO="lane:-1,prep:0,main:1,qual:1,coder:1" (an experiment's knob): the level (+1 highest, -1 lowest, 0 default) the streams of
// a role are made at, instead of the compressor's own choice (lanes +1, preparation -1, the others 0).  Returns `dflt` for a role not named.
static inline int cl_role_level(int role, int dflt)
{
	static const struct Cfg { bool on[CL_N_ROLES] = {}; int level[CL_N_ROLES] = {}; Cfg() {
		const char* e = getenv("COLORD_HIP_ROLE_PRIO"); if (!e) return;
		static const char* names[CL_N_ROLES] = { "main", "qual", "lane", "prep", "coder" };
		std::string t(e); size_t p = 0;
		while (p < t.size()) { size_t q = t.find(',', p); if (q == std::string::npos) q = t.size(); const std::string item = t.substr(p, q - p); const size_t c = item.find(':');
			if (c != std::string::npos) for (int r = 0; r < CL_N_ROLES; ++r) if (item.substr(0, c) == names[r]) { on[r] = true; level[r] = atoi(item.c_str() + c + 1); }
			p = q + 1; }
	} } cfg;
	return role >= 0 && role < CL_N_ROLES && cfg.on[role] ? cfg.level[role] : dflt;
}
static inline int cl_level_to_prio(int level)
{
	int least = 0, greatest = 0;
	if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); return 0; }
	return level > 0 ? greatest : level < 0 ? least : 0;
}
static inline hipError_t cl_stream_create_role(int role, int prio, hipStream_t* s)
{
	const CuMaskCfg& m = cl_cu_mask_cfg();
	if (role >= 0 && role < CL_N_ROLES && m.on[role])
	{
		uint32_t bits[32] = { 0 };
		for (uint32_t b = m.lo[role]; b <= m.hi[role]; ++b) bits[b >> 5] |= 1u << (b & 31);
		return hipExtStreamCreateWithCUMask(s, (m.hi[role] >> 5) + 1, bits);
	}
	return prio ? hipStreamCreateWithPriority(s, hipStreamNonBlocking, prio) : hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}
// a further stream of the context, at the context's priority (or on its role's CUs)
static inline hipError_t cl_stream_create(cl_ctx* c, hipStream_t* s) { return cl_stream_create_role(c->role, c->prio, s); }
// ... one of the context's side streams, created at first use (owner thread)
static inline hipError_t cl_side_stream(cl_ctx* c, std::atomic<hipStream_t>& slot)
{
	if (slot.load(std::memory_order_acquire)) return hipSuccess;
	hipStream_t s = nullptr;
	const hipError_t e = cl_stream_create(c, &s);
	if (e == hipSuccess) slot.store(s, std::memory_order_release);
	return e;
}
static inline void cl_timing_begin(cl_ctx*) {}   // times accumulate until cl_ctx_kernel_times reports them
// Adds what has COMPLETED to the kernel times (wait = true: everything; cl_ctx_kernel_times).  Never a wait by default: the events
// of an interval coder that is still running for the next batch (cl_dna_evolve_ahead) stay pending — waiting for them here, at
// the end of every stage, serialised the coders of consecutive chunks.
static inline void cl_timing_collect(cl_ctx* c, bool wait = false)
{
	if (!c->timing) return;
	size_t keep = 0;
	for (size_t i = 0; i < c->pending.size(); ++i)
	{
		auto& p = c->pending[i];
		bool done = true;
		if (wait) (void)hipEventSynchronize(p.second.second);
		else if (hipEventQuery(p.second.second) != hipSuccess) { (void)hipGetLastError(); done = false; }
		if (!done)
		{
			if (keep != i) { c->pending[keep] = std::move(p); c->pending_bytes[keep] = c->pending_bytes[i]; c->pending_cells[keep] = c->pending_cells[i]; }
			++keep;
			continue;
		}
		float ms = 0; (void)hipEventElapsedTime(&ms, p.second.first, p.second.second);
		std::string nm = p.first;
		if (!nm.empty() && nm.front() == '(' && nm.back() == ')') nm = nm.substr(1, nm.size() - 2);
		auto& t = c->times[nm]; t.ms += ms; t.launches += 1; t.bytes += c->pending_bytes[i]; t.cells += c->pending_cells[i];
		c->ev_pool.push_back(p.second.first); c->ev_pool.push_back(p.second.second);
	}
	c->pending.resize(keep); c->pending_bytes.resize(keep); c->pending_cells.resize(keep);
}

// ---- device primitives ---------------------------------------------------------------------------
// MurmurHash3 fmix64 — the reference's filter hash (filtering-KMC/hash_filter.h:8-16)
__host__ __device__ static inline uint64_t hash_mm(uint64_t x)
{
	x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
	x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
	x ^= x >> 33;
	return x;
}

// Exact "h % f == 0" for a launch-constant f without a divide: f = 2^s * d (d odd);
// h divisible by f  <=>  low s bits of h are zero  and  (h >> s) * inv(d) mod 2^64 <= (2^64-1)/d.
struct ModTest { uint64_t low_mask, dinv, lim; uint32_t s; };
static inline ModTest make_modtest(uint32_t f)
{
	ModTest m; uint32_t s = 0; uint64_t d = f;
	while ((d & 1) == 0) { d >>= 1; ++s; }
	uint64_t inv = d;                       // Newton iteration for the inverse of odd d modulo 2^64
	for (int i = 0; i < 6; ++i) inv *= 2 - d * inv;
	m.s = s; m.low_mask = (s ? ((1ULL << s) - 1) : 0); m.dinv = inv; m.lim = ~0ULL / d;
	return m;
}
__device__ static inline bool mod_is_zero(uint64_t h, const ModTest& m)
{
	return ((h & m.low_mask) == 0) && (((h >> m.s) * m.dinv) <= m.lim);
}

__device__ static inline uint32_t lane_id() { return threadIdx.x & 63; }
__device__ static inline uint64_t lanemask_lt() { return (1ULL << lane_id()) - 1; }

// inclusive wave scan (sum) over 64 lanes
__device__ static inline uint32_t wave_incl_scan(uint32_t v)
{
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(v, d, 64); if ((int)lane_id() >= d) v += t; }
	return v;
}
__device__ static inline uint32_t wave_sum(uint32_t v)
{
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
	return v;
}

// Block-wide exclusive scan for blockDim.x == 256 (4 waves).  sh must hold 4 uint32.  Returns the
// exclusive prefix of v; *total receives the block sum.
__device__ static inline uint32_t block_excl_scan_256(uint32_t v, uint32_t* sh, uint32_t* total)
{
	uint32_t incl = wave_incl_scan(v);
	uint32_t w = threadIdx.x >> 6;
	__syncthreads();                       // protect sh from a previous use
	if (lane_id() == 63) sh[w] = incl;
	__syncthreads();
	uint32_t base = 0, tot = 0;
#pragma unroll
	for (uint32_t i = 0; i < 4; ++i) { uint32_t s = sh[i]; if (i < w) base += s; tot += s; }
	*total = tot;
	return base + incl - v;
}

// device-wide primitives implemented in scan.hip / sort.hip
cl_status dev_exclusive_scan_u32(cl_ctx* ctx, uint32_t* d_data, uint64_t n, uint64_t* h_total);   // in place
cl_status dev_exclusive_scan_u64(cl_ctx* ctx, const uint32_t* d_in, uint64_t* d_out, uint64_t n, uint64_t* h_total); // d_out has n+1
cl_status dev_run_starts_u32(cl_ctx* ctx, const uint32_t* d_keys, uint64_t n, uint32_t shift, uint32_t* d_seg, uint64_t seg_cap, uint64_t* h_n_runs);   // scan.hip: starts of the runs of equal key >> shift in sorted keys
cl_status dev_run_starts_u64(cl_ctx* ctx, const uint64_t* d_keys, uint64_t n, uint32_t shift, uint32_t* d_seg, uint64_t seg_cap, uint64_t* h_n_runs);
cl_status dev_sort_pairs(cl_ctx* ctx, uint64_t* d_keys, uint32_t* d_vals, uint64_t n, uint32_t begin_bit, uint32_t end_bit);
cl_status dev_sort_pairs_swap(cl_ctx* ctx, DevBuf<uint64_t>& keys, DevBuf<uint32_t>& vals, uint64_t n, uint32_t begin_bit, uint32_t end_bit);   // buffers of exactly n elements may come back swapped with the sort's temporaries (no copy back)
cl_status dev_sort_keys32_pairs(cl_ctx* ctx, uint32_t* d_keys, uint32_t* d_vals, uint64_t n, uint32_t begin_bit, uint32_t end_bit);
cl_status dev_sort_keys32_pairs_swap(cl_ctx* ctx, DevBuf<uint32_t>& keys, DevBuf<uint32_t>& vals, uint64_t n, uint32_t begin_bit, uint32_t end_bit);   // as dev_sort_pairs_swap

static inline uint32_t grid_for(uint64_t n, uint32_t per_block) { return (uint32_t)((n + per_block - 1) / per_block); }
