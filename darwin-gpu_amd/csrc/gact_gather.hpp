// gact_gather.hpp -- the one collective of a sharded job for C / C++ callers: every rank's overlap records,
// narrowed on the device to the 32-byte printable line, gathered on rank 0 with RCCL, straight out of the engine's
// device-resident record array.  bench.py does the same through torch.distributed (gact_amd/dist.py RecordGather); this is
// the path of host/darwin_hip --shard R/W --rccl-gather and of any C caller of include/gact_hip.h.
//
// The reference has nothing to replace here (cuda_host.cu:195 is cudaSetDevice(0): one device); its way of joining the
// outputs of several processes is `cat darwin.*.out | sort | uniq` (README:25).
//
// RCCL is looked up when the first communicator is made (dlopen), not at link time: libgact_hip.so is loaded into
// processes that bring their own copy of it (torch bundles one), and a single-GPU caller needs none.
//
// Included by gact_engine.hip behind the engine's definitions (Slot, gact_hip_engine, fail, HIP_TRY).
// (system headers it needs -- dlfcn.h, fcntl.h, unistd.h, rccl/rccl.h for the types -- are included at the top of that file:
// this one sits inside its extern "C" block.)
#pragma once

namespace gact {

// gact_overlap -> gact_line (the eight numbers of gact.cpp:214-224; `emitted` folded into bit 1 of comp_emitted)
__global__ void pack_lines_kernel(const gact_overlap *__restrict__ rec, int n, gact_line *__restrict__ out)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const gact_overlap r = rec[k];
        gact_line l;
        l.ref_id = r.ref_id; l.query_id = r.query_id;
        l.ab = r.ab; l.ae = r.ae; l.bb = r.bb; l.be = r.be;
        l.score = r.score;
        l.comp_emitted = (r.comp & 1) | (r.emitted ? 2 : 0);
        out[k] = l;
    }
}

}  // namespace gact

namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

std::mutex g_rccl_mu;
RcclApi g_rccl;

// 0, or the engine's error code with the message set
int load_rccl()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return 0;
    // a copy that is in the process already wins (RTLD_NOLOAD): one RCCL per process
    const char *names[] = {opt_env("rccl_lib"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *lib = nullptr;
    for (const char *nm : names)
        if (nm && *nm && (lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) break;
    for (size_t k = 0; !lib && k < sizeof names / sizeof *names; k++)
        if (names[k] && *names[k]) lib = dlopen(names[k], RTLD_NOW | RTLD_LOCAL);
    if (!lib) return fail(GACT_HIP_EDEVICE, "RCCL not found (librccl.so.1; GACT_HIP_RCCL_LIB names another file): %s", dlerror());
    RcclApi a;
    a.lib = lib;
    bool ok = true;
    auto sym = [&](const char *nm) { void *p = dlsym(lib, nm); ok = ok && p != nullptr; return p; };
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
    a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
    a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) return fail(GACT_HIP_EDEVICE, "RCCL library lacks an entry point this engine calls");
    g_rccl = a;
    return 0;
}

#define RCCL_TRY(call)                                                                                       \
    do {                                                                                                     \
        ncclResult_t r_ = (call);                                                                            \
        if (r_ != ncclSuccess) return fail(GACT_HIP_EDEVICE, "%s: %s", #call, g_rccl.GetErrorString(r_));     \
    } while (0)

}  // namespace

struct gact_hip_comm {
    gact_hip_engine *e = nullptr;
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr;
    gact_line *d_lines = nullptr;        // this rank's lines
    size_t lines_cap = 0;
    gact_line *d_all = nullptr;          // rank 0: everybody's, in rank order
    size_t all_cap = 0;
    long long *d_counts = nullptr;       // [world + 1]: the ranks' counts, this rank's own behind them
    std::vector<long long> h_counts;
};

// rank 0 writes the id next to `path` and links it into place (a reader never sees half a file; a file that is there
// already is somebody else's: refused); the other ranks wait for it
static int exchange_id(int rank, const char *path, int timeout_s, ncclUniqueId *id)
{
    if (rank == 0) {
        RCCL_TRY(g_rccl.GetUniqueId(id));
        const std::string tmp = std::string(path) + ".tmp." + std::to_string((long)getpid());
        const int fd = open(tmp.c_str(), O_CREAT | O_EXCL | O_WRONLY, 0600);
        if (fd < 0) return fail(GACT_HIP_EINVAL, "comm_create: cannot create '%s'", tmp.c_str());
        const bool wrote = write(fd, id, sizeof *id) == (ssize_t)sizeof *id && fsync(fd) == 0;
        close(fd);
        bool linked = wrote && link(tmp.c_str(), path) == 0;
        const int link_errno = errno;
        // (a file system without hard links: look, then rename -- not atomic against another rank 0, but that is another job's
        //  mistake to make, not this one's)
        if (wrote && !linked && link_errno != EEXIST && access(path, F_OK) != 0) linked = rename(tmp.c_str(), path) == 0;
        unlink(tmp.c_str());
        if (!linked) return fail(GACT_HIP_EINVAL, "comm_create: '%s' exists already (an earlier job's?) or cannot be written", path);
        return 0;
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(timeout_s > 0 ? timeout_s : 120);
    for (;;) {
        const int fd = open(path, O_RDONLY);
        if (fd >= 0) {
            const ssize_t got = read(fd, id, sizeof *id);
            close(fd);
            if (got == (ssize_t)sizeof *id) return 0;
        }
        if (std::chrono::steady_clock::now() >= deadline)
            return fail(GACT_HIP_EDEVICE, "comm_create: rank %d waited %d s for rank 0's id in '%s'", rank, timeout_s > 0 ? timeout_s : 120, path);
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
}

int gact_hip_comm_create(gact_hip_engine *e, int32_t rank, int32_t world, const char *id_path, int32_t timeout_s, gact_hip_comm **out)
{
    if (!e || !out || !id_path || !*id_path) return fail(GACT_HIP_EINVAL, "comm_create: NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(GACT_HIP_EINVAL, "comm_create: rank %d of %d", rank, world);
    *out = nullptr;
    int rc = set_device(e);
    if (rc) return rc;
    if ((rc = load_rccl())) return rc;
    ncclUniqueId id;
    memset(&id, 0, sizeof id);
    if ((rc = exchange_id(rank, id_path, timeout_s, &id))) return rc;
    gact_hip_comm *c = new gact_hip_comm();
    c->e = e; c->rank = rank; c->world = world;
    c->h_counts.assign((size_t)world, 0);
    auto bail = [&](int code) { gact_hip_comm_destroy(c); return code; };
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipMalloc((void **)&c->d_counts, ((size_t)world + 1) * sizeof(long long)) != hipSuccess)
        return bail(fail(GACT_HIP_ENOMEM, "comm_create: stream / buffer creation failed"));
    {
        const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);      // collective: returns when every rank is in
        if (r != ncclSuccess) { c->comm = nullptr; return bail(fail(GACT_HIP_EDEVICE, "ncclCommInitRank: %s", g_rccl.GetErrorString(r))); }
    }
    if (rank == 0) unlink(id_path);           // everybody has read it
    *out = c;
    return 0;
}

int gact_hip_comm_destroy(gact_hip_comm *c)
{
    if (!c) return 0;
    if (c->e) (void)hipSetDevice(c->e->params.device_id);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    if (c->d_lines) (void)hipFree(c->d_lines);
    if (c->d_all) (void)hipFree(c->d_all);
    if (c->d_counts) (void)hipFree(c->d_counts);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int gact_hip_comm_gather_lines(gact_hip_comm *c, int slot, int32_t n, int64_t *counts, gact_line *lines, int64_t lines_cap)
{
    if (!c || !c->comm) return fail(GACT_HIP_EINVAL, "comm_gather_lines: no communicator");
    gact_hip_engine *e = c->e;
    int rc = check_slot(e, slot);
    if (rc) return rc;
    Slot &sl = e->slots[slot];
    if (n < 0 || (size_t)n > sl.n_cands) return fail(GACT_HIP_EINVAL, "comm_gather_lines: slot %d holds %zu candidates, not %d", slot, sl.n_cands, n);
    if ((rc = set_device(e))) return rc;
    // behind the slot's run: its records are complete in HBM when the collective reads them
    if (sl.stream) {
        HIP_TRY(hipEventRecord(c->ready, sl.stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ready, 0));
    }
    if ((size_t)n > c->lines_cap) {
        if (c->d_lines) (void)hipFree(c->d_lines);
        c->d_lines = nullptr; c->lines_cap = 0;
        const size_t want = std::max<size_t>((size_t)n, 4096);
        if (hipMalloc((void **)&c->d_lines, want * sizeof(gact_line)) != hipSuccess) return fail(GACT_HIP_ENOMEM, "comm_gather_lines: device allocation failed");
        c->lines_cap = want;
    }
    if (n > 0) {
        hipLaunchKernelGGL(gact::pack_lines_kernel, dim3(std::max(1, std::min((n + 255) / 256, 2048))), dim3(256), 0, c->stream, sl.overlaps.p, n, c->d_lines);
        HIP_TRY(hipGetLastError());
    }
    // the counts, once around (8 bytes per rank)
    const long long mine = n;
    HIP_TRY(hipMemcpyAsync(c->d_counts + c->world, &mine, sizeof mine, hipMemcpyHostToDevice, c->stream));
    RCCL_TRY(g_rccl.AllGather(c->d_counts + c->world, c->d_counts, 1, ncclInt64, c->comm, c->stream));
    HIP_TRY(hipMemcpyAsync(c->h_counts.data(), c->d_counts, (size_t)c->world * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    long long total = 0;
    for (int r = 0; r < c->world; r++) {
        if (c->h_counts[(size_t)r] < 0) return fail(GACT_HIP_EDEVICE, "comm_gather_lines: rank %d reports %lld records", r, c->h_counts[(size_t)r]);
        if (counts) counts[r] = c->h_counts[(size_t)r];
        total += c->h_counts[(size_t)r];
    }
    if (c->rank != 0) {
        // the gather itself: one send per rank that has records
        if (n > 0) RCCL_TRY(g_rccl.Send(c->d_lines, (size_t)n * sizeof(gact_line), ncclUint8, 0, c->comm, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return 0;
    }
    // (rank 0's caller may have brought too small an array, or none: the other ranks are in their sends by now and know
    //  nothing of it, so the receives are posted and waited for all the same -- into the communicator's own device array,
    //  sized from the counts -- and the error is reported once the collective is over.  `counts` holds what is needed:
    //  a call with lines == NULL is the way to ask.)
    const bool room = lines && lines_cap >= total;
    if ((size_t)total > c->all_cap) {
        if (c->d_all) (void)hipFree(c->d_all);
        c->d_all = nullptr; c->all_cap = 0;
        const size_t want = std::max<size_t>((size_t)total, 4096);
        if (hipMalloc((void **)&c->d_all, want * sizeof(gact_line)) != hipSuccess) return fail(GACT_HIP_ENOMEM, "comm_gather_lines: device allocation failed (%lld lines)", total);
        c->all_cap = want;
    }
    if (n > 0) HIP_TRY(hipMemcpyAsync(c->d_all, c->d_lines, (size_t)n * sizeof(gact_line), hipMemcpyDeviceToDevice, c->stream));
    RCCL_TRY(g_rccl.GroupStart());
    long long at = c->h_counts[0];
    for (int r = 1; r < c->world; r++) {
        const long long cnt = c->h_counts[(size_t)r];
        if (cnt > 0) {
            const ncclResult_t rr = g_rccl.Recv(c->d_all + at, (size_t)cnt * sizeof(gact_line), ncclUint8, r, c->comm, c->stream);
            if (rr != ncclSuccess) { (void)g_rccl.GroupEnd(); return fail(GACT_HIP_EDEVICE, "ncclRecv from rank %d: %s", r, g_rccl.GetErrorString(rr)); }
        }
        at += cnt;
    }
    RCCL_TRY(g_rccl.GroupEnd());
    if (room && total > 0) HIP_TRY(hipMemcpyAsync(lines, c->d_all, (size_t)total * sizeof(gact_line), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!room) return fail(GACT_HIP_EINVAL, "comm_gather_lines: rank 0 needs room for %lld lines (got %lld); the gather itself was completed", total, (long long)(lines ? lines_cap : 0));
    return 0;
}
