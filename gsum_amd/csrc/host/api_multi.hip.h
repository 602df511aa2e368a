// C ABI: a group of contexts, one per GPU of the node, driven from ONE caller thread -- gsum_init_multi, gsum_group_*, gsum_lml_batch_multi
// (part of gsum_capi.hip: included from there, in order -- one translation unit)
//
// The reference's user is one Python process (docs/notebooks/correlated_EFT_publication.ipynb:1444-1459; gsum/models.py:1485-1507): the
// likelihood scan it loops over is a pure map, so a group cuts the descriptor list with gsum_shard_range, runs every device's block on a
// host thread of its own (a gsum_ctx is bound to one GPU and to one thread at a time; no data-path collective) and writes the results
// straight into the caller's full-length arrays.  The exchange step the path has -- the gather of the log-likelihood grid -- is
// optional on top: an in-place ncclAllGather over device-resident copies of the result arrays (RCCL over xGMI; librccl is loaded on
// first use, the library does not link it).
#pragma once

#include <dlfcn.h>
#include <thread>

struct gs_rccl_api {
    void* handle = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*AllGather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
enum { GS_NCCL_DOUBLE = 8 };          // ncclDouble == ncclFloat64 (rccl/rccl.h)

struct gsum_group {
    std::vector<gsum_ctx*> ctx;
    std::vector<int> dev;               // their devices (kept apart: an adopted context may be gone by the time the group is destroyed)
    bool owns = true;                   // gsum_init_multi: the group made its contexts; gsum_group_adopt: the caller did and keeps them
    std::string err;
    gs_rccl_api rccl;
    int rccl_state = 0;                 // 0 not tried, 1 communicators open, -1 unavailable (err says why)
    std::vector<void*> comm;
    std::vector<double*> gbuf;          // per device: the gather buffer (world x chunk rows)
    std::vector<size_t> gcap;
    int64_t gathers = 0;                // RCCL all-gathers run so far (gsum_group_get "rccl_gathers")
    int last_world_used = 0;            // devices that had work in the last sharded call
};

static std::string g_group_init_error;

int gsum_init_multi(int n_devices, const int* device_ids, gsum_group** out) {
    if (!out) return -2;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_group_init_error = std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "count = 0");
        return -1;
    }
    if (n_devices <= 0) {               // every visible GPU
        n_devices = count;
        device_ids = nullptr;
    }
    if (n_devices > 64) {
        g_group_init_error = "gsum_init_multi: more than 64 devices";
        return -2;
    }
    gsum_group* g = new gsum_group();
    for (int i = 0; i < n_devices; ++i) {
        const int dev = device_ids ? device_ids[i] : i;
        gsum_ctx* c = nullptr;
        const int rc = gsum_init(dev, &c);          // creates the context's four streams back to back (four pipes), per device
        if (rc) {
            g_group_init_error = "gsum_init_multi: device " + std::to_string(dev) + ": " + g_init_error;
            for (gsum_ctx* p : g->ctx) gsum_destroy(p);
            delete g;
            return rc;
        }
        g->ctx.push_back(c);
        g->dev.push_back(dev);
    }
    g->gbuf.assign(n_devices, nullptr);
    g->gcap.assign(n_devices, 0);
    *out = g;
    return 0;
}

int gsum_group_adopt(int n_ctx, gsum_ctx* const* ctxs, gsum_group** out) {
    if (!out) return -2;
    *out = nullptr;
    if (n_ctx < 1 || n_ctx > 64 || !ctxs) {
        g_group_init_error = "gsum_group_adopt: need 1..64 contexts";
        return -2;
    }
    for (int i = 0; i < n_ctx; ++i)
        for (int j = 0; j < i; ++j)
            if (!ctxs[i] || ctxs[i] == ctxs[j]) {
                g_group_init_error = "gsum_group_adopt: null or repeated context";
                return -2;
            }
    gsum_group* g = new gsum_group();
    g->owns = false;
    g->ctx.assign(ctxs, ctxs + n_ctx);
    for (int i = 0; i < n_ctx; ++i) g->dev.push_back(ctxs[i]->device);
    g->gbuf.assign(n_ctx, nullptr);
    g->gcap.assign(n_ctx, 0);
    *out = g;
    return 0;
}

void gsum_group_destroy(gsum_group* g) {
    if (!g) return;
    for (size_t r = 0; r < g->ctx.size(); ++r) {
        (void)hipSetDevice(g->dev[r]);
        (void)hipDeviceSynchronize();
        if (r < g->comm.size() && g->comm[r] && g->rccl.CommDestroy) (void)g->rccl.CommDestroy(g->comm[r]);
        if (g->gbuf[r]) (void)hipFree(g->gbuf[r]);
    }
    if (g->owns)
        for (gsum_ctx* c : g->ctx) gsum_destroy(c);
    // librccl stays loaded: unloading it under a live HIP runtime is not safe
    delete g;
}

int32_t gsum_group_size(const gsum_group* g) { return g ? (int32_t)g->ctx.size() : 0; }

gsum_ctx* gsum_group_ctx(gsum_group* g, int32_t i) { return (g && i >= 0 && i < (int32_t)g->ctx.size()) ? g->ctx[i] : nullptr; }

const char* gsum_group_last_error(gsum_group* g) { return g ? g->err.c_str() : g_group_init_error.c_str(); }

int64_t gsum_group_get(gsum_group* g, const char* name) {
    if (!g || !name) return -1;
    if (!strcmp(name, "rccl")) return g->rccl_state;
    if (!strcmp(name, "rccl_gathers")) return g->gathers;
    if (!strcmp(name, "devices_used")) return g->last_world_used;
    return -1;
}

// run fn(rank) on one host thread per device (rank 0 on the caller's); the first failure's message goes to g->err
template <class F>
static int gs_group_run(gsum_group* g, F fn) {
    const int world = (int)g->ctx.size();
    std::vector<int> rc(world, 0);
    std::vector<std::thread> th;
    th.reserve(world);
    for (int r = 1; r < world; ++r) th.emplace_back([&rc, &fn, r]() { rc[r] = fn(r); });
    rc[0] = fn(0);
    for (auto& t : th) t.join();
    for (int r = 0; r < world; ++r)
        if (rc[r]) {
            g->err = "device " + std::to_string(g->ctx[r]->device) + " (rank " + std::to_string(r) + "): " + g->ctx[r]->err;
            return rc[r];
        }
    return 0;
}

int gsum_group_set_inputs(gsum_group* g, const double* X, int64_t n, int32_t d, const double* RHS, int32_t k) {
    if (!g) return -2;
    return gs_group_run(g, [&](int r) { return gsum_set_inputs(g->ctx[r], X, n, d, RHS, k); });
}

// every device keeps ALL right-hand-side sets (a surface's 64 sets at n = 8192, k = 7: 29 MB): the descriptors, not the sets, are partitioned
int gsum_group_set_inputs_sets(gsum_group* g, const double* X, int64_t n, int32_t d, const double* RHS_sets, int32_t n_sets, int32_t k) {
    if (!g) return -2;
    return gs_group_run(g, [&](int r) { return gsum_set_inputs_sets(g->ctx[r], X, n, d, RHS_sets, n_sets, k); });
}

// ---- RCCL (optional): communicators over the group's devices, opened on first use ---------------------------------------------------
static int gs_group_rccl_open(gsum_group* g) {
    if (g->rccl_state == 1) return 0;
    if (g->rccl_state < 0) return -1;                       // g->err still says why
    g->rccl_state = -1;
    const int world = (int)g->ctx.size();
    std::vector<int> devs(world);
    for (int r = 0; r < world; ++r) devs[r] = g->ctx[r]->device;
    for (int a = 0; a < world; ++a)
        for (int b = a + 1; b < world; ++b)
            if (devs[a] == devs[b]) {
                g->err = "RCCL gather: the group lists device " + std::to_string(devs[a]) + " twice (one rank per GPU)";
                return -1;
            }
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        const char* why = dlerror();
        g->err = std::string("RCCL gather: librccl.so not loadable: ") + (why ? why : "?");
        return -1;
    }
    gs_rccl_api& A = g->rccl;
    A.handle = h;
    A.CommInitAll = (decltype(A.CommInitAll))dlsym(h, "ncclCommInitAll");
    A.CommDestroy = (decltype(A.CommDestroy))dlsym(h, "ncclCommDestroy");
    A.AllGather = (decltype(A.AllGather))dlsym(h, "ncclAllGather");
    A.GroupStart = (decltype(A.GroupStart))dlsym(h, "ncclGroupStart");
    A.GroupEnd = (decltype(A.GroupEnd))dlsym(h, "ncclGroupEnd");
    A.GetErrorString = (decltype(A.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!A.CommInitAll || !A.CommDestroy || !A.AllGather || !A.GroupStart || !A.GroupEnd || !A.GetErrorString) {
        g->err = "RCCL gather: librccl.so lacks an expected symbol";
        return -1;
    }
    g->comm.assign(world, nullptr);
    const int rc = A.CommInitAll(g->comm.data(), world, devs.data());
    if (rc) {
        g->err = std::string("ncclCommInitAll: ") + A.GetErrorString(rc);
        g->comm.clear();
        return -1;
    }
    g->rccl_state = 1;
    return 0;
}

// Gather a row-partitioned host array through the devices: rank r's rows [lo_r, hi_r) (gsum_shard_range) go to ITS device's buffer at
// their positions, ONE in-place ncclAllGather per device exchanges the blocks over xGMI, device 0's copy comes back into `buf` and
// every other device's copy is compared with it byte for byte (a gather that disagrees between ranks is an error, not a result).
int gsum_group_allgather(gsum_group* g, double* buf, int64_t rows, int64_t width) {
    if (!g || !buf || rows < 0 || width < 1) return -2;
    if (gs_group_rccl_open(g)) return -1;
    const int world = (int)g->ctx.size();
    if (rows == 0) return 0;
    const int64_t chunk = (rows + world - 1) / world;
    const size_t bytes = (size_t)world * chunk * width * sizeof(double);
    auto hipfail = [&](const char* what, hipError_t e) {
        g->err = std::string(what) + ": " + hipGetErrorString(e);
        return -1;
    };
    hipError_t e;
    for (int r = 0; r < world; ++r) {
        gsum_ctx* c = g->ctx[r];
        if ((e = hipSetDevice(c->device)) != hipSuccess) return hipfail("hipSetDevice", e);
        if (g->gcap[r] < bytes) {
            if (g->gbuf[r]) (void)hipFree(g->gbuf[r]);
            g->gbuf[r] = nullptr;
            g->gcap[r] = 0;
            if ((e = hipMalloc((void**)&g->gbuf[r], bytes)) != hipSuccess) return hipfail("hipMalloc (gather buffer)", e);
            g->gcap[r] = bytes;
        }
        int64_t lo = 0, hi = 0;
        (void)gsum_shard_range(rows, r, world, &lo, &hi);
        // only this rank's block is staged: whatever the rest of its buffer holds is overwritten by the collective
        if (hi > lo && (e = hipMemcpyAsync(g->gbuf[r] + lo * width, buf + lo * width, (size_t)(hi - lo) * width * sizeof(double),
                                           hipMemcpyHostToDevice, c->slots[0].sm)) != hipSuccess)
            return hipfail("hipMemcpyAsync (stage slice)", e);
    }
    int rc = g->rccl.GroupStart();
    for (int r = 0; r < world && !rc; ++r)
        rc = g->rccl.AllGather(g->gbuf[r] + (size_t)r * chunk * width, g->gbuf[r], (size_t)chunk * width, GS_NCCL_DOUBLE, g->comm[r],
                               g->ctx[r]->slots[0].sm);
    const int rc_end = g->rccl.GroupEnd();
    if (rc || rc_end) {
        g->err = std::string("ncclAllGather: ") + g->rccl.GetErrorString(rc ? rc : rc_end);
        return -1;
    }
    ++g->gathers;
    std::vector<double> other;
    for (int r = 0; r < world; ++r) {
        gsum_ctx* c = g->ctx[r];
        if ((e = hipSetDevice(c->device)) != hipSuccess) return hipfail("hipSetDevice", e);
        double* dst = buf;
        if (r > 0) {
            other.resize((size_t)rows * width);
            dst = other.data();
        }
        if ((e = hipMemcpyAsync(dst, g->gbuf[r], (size_t)rows * width * sizeof(double), hipMemcpyDeviceToHost, c->slots[0].sm)) != hipSuccess)
            return hipfail("hipMemcpyAsync (read back)", e);
        if ((e = hipStreamSynchronize(c->slots[0].sm)) != hipSuccess) return hipfail("hipStreamSynchronize", e);
        if (r > 0 && memcmp(dst, buf, (size_t)rows * width * sizeof(double)) != 0) {
            g->err = "RCCL gather: rank " + std::to_string(r) + " holds a different grid than rank 0";
            return -1;
        }
    }
    return 0;
}

// ---- the sharded scan ------------------------------------------------------------------------------------------------------------------
static int gs_group_lml(gsum_group* g, bool resident, const gsum_kernel_desc* kernels, int32_t n_kernels, double nugget, double* G_out,
                        double* sld_out, int64_t* info_out, int32_t flags, const int32_t* set_of = nullptr) {
    if (!kernels || !G_out || !sld_out || !info_out || n_kernels < 0) {
        g->err = "gsum_lml_batch_multi: null argument";
        return -2;
    }
    if (flags & ~GSUM_GATHER_RCCL) {
        g->err = "gsum_lml_batch_multi: unknown flag";
        return -2;
    }
    if ((flags & GSUM_GATHER_RCCL) && gs_group_rccl_open(g)) return -1;     // before any work is spent
    const int world = (int)g->ctx.size();
    int used = 0;
    for (int r = 0; r < world; ++r) {
        int64_t lo = 0, hi = 0;
        (void)gsum_shard_range(n_kernels, r, world, &lo, &hi);
        used += hi > lo;
    }
    g->last_world_used = used;
    const int rc = gs_group_run(g, [&](int r) {
        gsum_ctx* c = g->ctx[r];
        int64_t lo = 0, hi = 0;
        (void)gsum_shard_range(n_kernels, r, world, &lo, &hi);
        if (hi == lo) return 0;
        gs_inputs* I = resident ? &c->res : &c->op;
        const int64_t kk = (int64_t)I->k * I->k;
        return gs_lml_on(c, I, kernels + lo, (int32_t)(hi - lo), nugget, G_out + lo * kk, sld_out + lo, info_out + lo, set_of ? set_of + lo : nullptr);
    });
    if (rc || !(flags & GSUM_GATHER_RCCL) || n_kernels == 0) return rc;
    // the grid's exchange step on the devices: sld, G and info (exact as fp64: LAPACK's info <= n < 2^53) as ONE packed array
    const int k = resident ? g->ctx[0]->res.k : g->ctx[0]->op.k;
    const int64_t kk = (int64_t)k * k, W = kk + 2;
    std::vector<double> packed((size_t)n_kernels * W);
    for (int64_t i = 0; i < n_kernels; ++i) {
        double* p = packed.data() + i * W;
        memcpy(p, G_out + i * kk, (size_t)kk * sizeof(double));
        p[kk] = sld_out[i];
        p[kk + 1] = (double)info_out[i];
    }
    if (gsum_group_allgather(g, packed.data(), n_kernels, W)) return -1;
    for (int64_t i = 0; i < n_kernels; ++i) {
        const double* p = packed.data() + i * W;
        memcpy(G_out + i * kk, p, (size_t)kk * sizeof(double));
        sld_out[i] = p[kk];
        info_out[i] = (int64_t)p[kk + 1];
    }
    return 0;
}

int gsum_group_lml_resident(gsum_group* g, const gsum_kernel_desc* kernels, int32_t n_kernels, double nugget, double* G_out,
                            double* sld_out, int64_t* info_out, int32_t flags) {
    if (!g) return -2;
    return gs_group_lml(g, true, kernels, n_kernels, nugget, G_out, sld_out, info_out, flags);
}

int gsum_group_lml_resident_sets(gsum_group* g, const gsum_kernel_desc* kernels, const int32_t* set_of, int32_t n_kernels, double nugget,
                                 double* G_out, double* sld_out, int64_t* info_out, int32_t flags) {
    if (!g) return -2;
    return gs_group_lml(g, true, kernels, n_kernels, nugget, G_out, sld_out, info_out, flags, set_of);
}

int gsum_lml_batch_multi(gsum_group* g, const gsum_kernel_desc* kernels, int32_t n_kernels, const double* X, int64_t n, int32_t d,
                         const double* RHS, int32_t k, double nugget, double* G_out, double* sld_out, int64_t* info_out, int32_t flags) {
    if (!g) return -2;
    const int rc = gs_group_run(g, [&](int r) { return gs_upload_inputs(g->ctx[r], X, n, d, RHS, k); });
    if (rc) return rc;
    return gs_group_lml(g, false, kernels, n_kernels, nugget, G_out, sld_out, info_out, flags);
}
