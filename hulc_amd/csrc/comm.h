// hulc_amd/csrc/comm.h — the gradient all-reduce of the data-parallel step, owned by the library: RCCL over xGMI on its own HIP stream.
//
// Replaces the reference's Lightning DDPStrategy (hulc/training.py:64-69: DDP(find_unused_parameters=False) = mean of per-rank gradients);
// the 1/world factor stays folded into the Adam kernel (grad_scale).  One process per GPU; the host passes in the 128-byte ncclUniqueId
// it distributed among the ranks (torch.distributed / any store), this file does the rest:
//   * RCCL is resolved at run time (dlopen "librccl.so.1": in a PyTorch-ROCm process that is the copy torch already loaded, so there is ONE
//     RCCL instance per process; nothing links against it and the library still loads on a box without RCCL),
//   * collectives run on a private high-priority stream; an event from the engine stream gates each bucket, one event back gates whatever
//     the engine enqueues after the backward (Adam) — no host synchronisation anywhere,
//   * buckets are module groups of the flat gradient buffer in REVERSE-FORWARD order (action decoder -> plan proposal -> plan recognition ->
//     goal encoders -> perceptual encoders): each is reduced as soon as the backward stage that finalises it has been enqueued, so the
//     first 61 MB leave while the rest of the backward still computes (Engine::backward, engine.h).
//   * bucket dtype fp32 (the reference's: fp32 gradients) or bf16 (half the bytes on the wire: xGMI rings are per-link bound; the sum is
//     then taken in bf16 by RCCL and widened back, a precision trade the host opts into).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

void hulc_set_error(const char* fmt, ...);

struct GradComm {
    typedef struct { char internal[128]; } UniqueId;          // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
    typedef void* Comm;
    enum { F16 = 6, F32 = 7, BF16 = 9, SUM = 0 };             // ncclDataType_t / ncclRedOp_t values (rccl.h)
    typedef int (*GetUniqueIdFn)(UniqueId*);
    typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
    typedef int (*CommDestroyFn)(Comm);
    typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
    typedef const char* (*GetErrorStringFn)(int);
    typedef int (*CommQueryFn)(const Comm, int*);             // ncclCommCount / ncclCommUserRank
    struct Api { void* lib = nullptr; GetUniqueIdFn get_id = nullptr; CommInitRankFn init = nullptr; CommDestroyFn destroy = nullptr; AllReduceFn allreduce = nullptr; GetErrorStringFn errstr = nullptr;
                 CommQueryFn count = nullptr, user_rank = nullptr; };

    static Api& api() { static Api a; return a; }
    static bool load_api() {
        Api& a = api();
        if (a.lib) return true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) { a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (a.lib) break; }
        if (!a.lib) { hulc_set_error("RCCL not found (dlopen librccl.so.1): %s", dlerror()); return false; }
        a.get_id = (GetUniqueIdFn)dlsym(a.lib, "ncclGetUniqueId"); a.init = (CommInitRankFn)dlsym(a.lib, "ncclCommInitRank");
        a.destroy = (CommDestroyFn)dlsym(a.lib, "ncclCommDestroy"); a.allreduce = (AllReduceFn)dlsym(a.lib, "ncclAllReduce");
        a.errstr = (GetErrorStringFn)dlsym(a.lib, "ncclGetErrorString");
        a.count = (CommQueryFn)dlsym(a.lib, "ncclCommCount"); a.user_rank = (CommQueryFn)dlsym(a.lib, "ncclCommUserRank");      // optional: hulc_comm_size
        if (!a.get_id || !a.init || !a.destroy || !a.allreduce) { hulc_set_error("RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce"); a.lib = nullptr; return false; }
        return true;
    }
    static const char* err(int rc) { return api().errstr ? api().errstr(rc) : "rccl error"; }

    Comm comm = nullptr;
    hipStream_t cs = nullptr;                                  // the collectives' stream
    int rank = 0, world = 1;
    std::vector<hipEvent_t> ev; size_t ev_used = 0;
    void* stage = nullptr; int64_t stage_elems = 0;            // bf16 staging buffer (bf16 bucket mode)
    long long n_collectives = 0; double bytes_reduced = 0;     // statistics (tests / bench JSON)
    // "comm_timing" (hulc_set_option): timing events around every bucket's collective of the LAST hulc_backward_allreduce, on the collectives'
    // stream, plus the begin / end of that backward on the engine stream -> hulc_comm_timeline: how much of each bucket hid under the backward
    struct Span { hipEvent_t t0 = nullptr, t1 = nullptr; double bytes = 0; bool used = false; };
    Span span[8];
    hipEvent_t bwd_t0 = nullptr, bwd_t1 = nullptr;
    int n_span = 0;
    bool timeline_valid = false;
    void span_begin(int i, double bytes) {
        if (i < 0 || i >= 8) return;
        Span& s = span[i];
        if (!s.t0) { hipEventCreate(&s.t0); hipEventCreate(&s.t1); }
        s.bytes = bytes; s.used = true;
        hipEventRecord(s.t0, cs);
        if (i + 1 > n_span) n_span = i + 1;
    }
    void span_end(int i) { if (i >= 0 && i < 8 && span[i].used) hipEventRecord(span[i].t1, cs); }
    void bwd_mark(bool begin, hipStream_t st) {
        if (!bwd_t0) { hipEventCreate(&bwd_t0); hipEventCreate(&bwd_t1); }
        if (begin) { for (Span& s : span) s.used = false; n_span = 0; timeline_valid = false; hipEventRecord(bwd_t0, st); }
        else { hipEventRecord(bwd_t1, st); timeline_valid = true; }
    }
    // out[4 i .. 4 i + 3] = {collective start, collective end (us, relative to the END of the backward on the engine stream: negative = hidden
    // under it), bytes on the wire per rank, 0}; *bwd_us = duration of the backward.  Synchronises both streams.  Returns the bucket count, < 0 on error
    int timeline(double* out, int cap, double* bwd_us, hipStream_t st) {
        if (!timeline_valid) { hulc_set_error("hulc_comm_timeline: no timed hulc_backward_allreduce yet (hulc_set_option comm_timing 1)"); return -1; }
        hipStreamSynchronize(cs); hipStreamSynchronize(st);
        float ms = 0;
        hipEventElapsedTime(&ms, bwd_t0, bwd_t1);
        if (bwd_us) *bwd_us = ms * 1e3;
        int n = 0;
        for (int i = 0; i < n_span && n < cap; ++i) {
            if (!span[i].used) continue;
            float a = 0, b = 0;
            hipEventElapsedTime(&a, bwd_t1, span[i].t0); hipEventElapsedTime(&b, bwd_t1, span[i].t1);
            out[4 * n] = a * 1e3; out[4 * n + 1] = b * 1e3; out[4 * n + 2] = span[i].bytes; out[4 * n + 3] = (double)i;
            ++n;
        }
        return n;
    }

    // everything that can fail on ONE rank only (RCCL not loadable, no stream): done before any rank enters the blocking ncclCommInitRank, so
    // the host can agree on the outcome first (hulc_comm_prepare -> all ranks vote -> hulc_comm_init) instead of deadlocking the healthy ranks
    int prepare() {
        if (!load_api()) return 1;
        if (cs) return 0;
        int lo = 0, hi = 0;
        hipDeviceGetStreamPriorityRange(&lo, &hi);             // hi = greatest priority (numerically lowest)
        if (hipStreamCreateWithPriority(&cs, hipStreamNonBlocking, hi) != hipSuccess) { cs = nullptr; hulc_set_error("hulc_comm_prepare: stream creation failed"); return 1; }
        return 0;
    }
    int init(const void* unique_id, int rank_, int world_) {
        if (prepare()) return 1;
        rank = rank_; world = world_;
        UniqueId id; memcpy(&id, unique_id, sizeof(id));
        const int rc = api().init(&comm, world, id, rank);
        if (rc != 0) { hulc_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, err(rc)); comm = nullptr; return 1; }
        return 0;
    }
    // what the LIVE communicator says about itself (ncclCommCount / ncclCommUserRank), not what the host passed to init(): the evidence bench.py
    // prints as allreduce.rccl_ranks and gates its N > 1 line on
    int size(int* rank_out, int* world_out) const {
        if (!comm) { hulc_set_error("hulc_comm_size: no communicator"); return 1; }
        int w = -1, r = -1;
        if (!api().count || !api().user_rank) { hulc_set_error("hulc_comm_size: this RCCL lacks ncclCommCount / ncclCommUserRank"); return 1; }
        const int rc1 = api().count(comm, &w), rc2 = api().user_rank(comm, &r);
        if (rc1 != 0 || rc2 != 0) { hulc_set_error("ncclCommCount / ncclCommUserRank failed: %s", err(rc1 ? rc1 : rc2)); return 1; }
        if (rank_out) *rank_out = r;
        if (world_out) *world_out = w;
        return 0;
    }
    ~GradComm() {
        if (comm) { hipStreamSynchronize(cs); api().destroy(comm); }
        for (hipEvent_t e : ev) hipEventDestroy(e);
        for (Span& s : span) { if (s.t0) hipEventDestroy(s.t0); if (s.t1) hipEventDestroy(s.t1); }
        if (bwd_t0) { hipEventDestroy(bwd_t0); hipEventDestroy(bwd_t1); }
        if (cs) hipStreamDestroy(cs);
        if (stage) hipFree(stage);
    }
    hipEvent_t next_event() {
        if (ev_used == ev.size()) { hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming); ev.push_back(e); }
        return ev[ev_used++];
    }
    // the collectives' stream waits for everything enqueued on `st` so far
    void gate_from(hipStream_t st) { hipEvent_t e = next_event(); hipEventRecord(e, st); hipStreamWaitEvent(cs, e, 0); }
    // `st` waits for everything enqueued on the collectives' stream so far; event slots are recycled per step
    void gate_to(hipStream_t st) { hipEvent_t e = next_event(); hipEventRecord(e, cs); hipStreamWaitEvent(st, e, 0); ev_used = 0; }
};
