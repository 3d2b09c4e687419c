// comm.hip -- the image-tile assembly of a sharded frame through RCCL (SURVEY.md section 8(b) "thin RCCL helpers: comm init from
// unique-id, all-gather"; section 8(e)).  The path shards embarrassingly: every rank renders a contiguous block of image rows and the ONLY
// exchange is one all-gather of the [rows_local * W, C] fp32 tiles per frame.  The reference is single-GPU (main.py:166-170): no counterpart.
//
// librccl is resolved at FIRST USE (dlopen), not at link time: a one-GPU user of libmi_nerf.so needs no RCCL at all, and inside a PyTorch
// process the library already mapped by torch (soname librccl.so.1) is the one picked up, so ProcessGroupNCCL and these helpers share one
// RCCL and one HIP runtime.  The collective is enqueued on the hipStream_t the caller passes -- the stream the render ran on -- so no host
// synchronisation or cross-stream event separates the last composite launch from the gather.
#include <dlfcn.h>
#include <mutex>
#include <set>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "common.h"

namespace minerf {

// the handful of RCCL declarations used (rccl.h: ncclUniqueId is 128 opaque bytes passed BY VALUE; ncclFloat32 = 7; ncclSuccess = 0)
struct RcclId { char internal[MI_NERF_COMM_ID_BYTES]; };
typedef void* RcclComm;
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(RcclComm*, int, RcclId, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    char why[256] = "";
};
static Rccl g_rccl;
static std::once_flag g_rccl_once;

static void load_rccl() {
    Rccl& r = g_rccl;
    const char* env = getenv("MI_NERF_RCCL_LIB");
    void* h = nullptr;
    if (env && *env) {
        h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    } else {
        h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);      // the one the process already has (PyTorch's)
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);            // else through this library's RUNPATH (/opt/rocm/lib)
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) {
        const char* e = dlerror();
        snprintf(r.why, sizeof(r.why), "%s", e ? e : "dlopen failed");
        return;
    }
    r.GetUniqueId = (int (*)(RcclId*))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(RcclComm*, int, RcclId, int))dlsym(h, "ncclCommInitRank");
    r.AllGather = (int (*)(const void*, void*, size_t, int, RcclComm, hipStream_t))dlsym(h, "ncclAllGather");
    r.CommDestroy = (int (*)(RcclComm))dlsym(h, "ncclCommDestroy");
    r.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy || !r.GetErrorString) {
        snprintf(r.why, sizeof(r.why), "the library lacks one of ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy / ncclGetErrorString");
        return;
    }
    // One HIP runtime per process: a librccl that came in through the fall-back dlopen may bind a libamdhip64 other than the one this library's
    // launches go to (a torch wheel ships its own); streams and device pointers of one runtime mean nothing to the other.  Refuse it.
    // (MI_NERF_RCCL_LIB names a library the caller vouches for -- the tests' stand-in links no second runtime.)
    if (!(env && *env)) {
        Dl_info mine{}, theirs{};
        void* their_sym = dlsym(h, "hipGetDevice");          // searched in librccl's own dependency order
        if (their_sym && dladdr(their_sym, &theirs) && dladdr((void*)&hipGetDevice, &mine) && mine.dli_fbase != theirs.dli_fbase) {
            snprintf(r.why, sizeof(r.why), "librccl binds another HIP runtime (%.80s) than this library (%.80s)",
                     theirs.dli_fname ? theirs.dli_fname : "?", mine.dli_fname ? mine.dli_fname : "?");
            return;
        }
    }
    r.handle = h;
}
static int rccl(const Rccl** out) {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.handle) {
        set_error("RCCL is not available (%s): the tile gather needs ROCm's librccl.so.1 on the loader path, or MI_NERF_RCCL_LIB=/path/to/librccl.so", g_rccl.why);
        return MI_NERF_ERCCL;
    }
    *out = &g_rccl;
    return MI_NERF_OK;
}
static int rccl_fail(const Rccl* r, int rc, const char* what) {
    set_error("RCCL error %d (%s) in %s", rc, r->GetErrorString(rc), what);
    return MI_NERF_ERCCL;
}

struct TileComm {
    RcclComm comm;
    int world, rank, device;
};
// live handles: a handle is valid iff it is in this set (no magic word read through a pointer that may have been freed)
static std::mutex g_comms_mu;
static std::set<TileComm*> g_comms;

// rows of rank r when H rows are split into `world` contiguous blocks: the first H % world ranks get one extra row (dist.shard_rows)
__host__ __device__ inline int block_rows(int H, int world, int r) { return H / world + (r < H % world ? 1 : 0); }
__host__ __device__ inline int block_row0(int H, int world, int r) { return r * (H / world) + (r < H % world ? r : H % world); }

// staging [world][max_rows * row_floats] (every rank's tile padded to the largest block) -> frame [H * row_floats].  VEC floats per thread.
template <int VEC>
__global__ __launch_bounds__(256) void unpad_tiles_kernel(const float* __restrict__ staging, int world, int H, long long row_vecs,
                                                          float* __restrict__ frame) {
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const long long total = (long long)H * row_vecs;
    const int base = H / world, extra = H % world;
    const long long max_vecs = (long long)(base + (extra ? 1 : 0)) * row_vecs;      // the largest block
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / row_vecs);
        const long long col = i - (long long)row * row_vecs;
        const int split = (base + 1) * extra;                                  // first row of the blocks without the extra row
        const int r = row < split ? row / (base + 1) : extra + (row - split) / base;
        const int local = row - block_row0(H, world, r);
        ((vec_t*)frame)[i] = ((const vec_t*)staging)[(long long)r * max_vecs + (long long)local * row_vecs + col];
    }
}

static int check_geometry(int world, int H, int W, int C) {
    MN_CHECK_ARG(world >= 1, "world = %d", world);
    MN_CHECK_ARG(H >= 1 && W >= 1 && C >= 1, "frame geometry H=%d W=%d C=%d", H, W, C);
    MN_CHECK_ARG(H >= world, "H = %d rows cannot be split over %d ranks (a rank would own no row)", H, world);
    MN_CHECK_ARG((long long)H * W * C < (1ll << 40), "frame of %d x %d x %d floats is out of range", H, W, C);
    return MI_NERF_OK;
}

int unpad_tiles(const float* staging_dev, int world, int H, int W, int C, float* frame_dev, hipStream_t stream) {
    if (int rc = check_geometry(world, H, W, C)) return rc;
    MN_CHECK_ARG(staging_dev && frame_dev, "staging / frame is NULL");
    const long long row_floats = (long long)W * C;
    const bool v4 = row_floats % 4 == 0 && ((uintptr_t)staging_dev % 16 == 0) && ((uintptr_t)frame_dev % 16 == 0);
    const long long row_vecs = v4 ? row_floats / 4 : row_floats;
    const long long total = (long long)H * row_vecs;
    const int blocks = (int)std::min<long long>((total + 255) / 256, (long long)device_cus() * 8);
    if (v4) unpad_tiles_kernel<4><<<blocks, 256, 0, stream>>>(staging_dev, world, H, row_vecs, frame_dev);
    else unpad_tiles_kernel<1><<<blocks, 256, 0, stream>>>(staging_dev, world, H, row_vecs, frame_dev);
    MN_LAUNCH_CHECK("unpad_tiles_kernel");
    return MI_NERF_OK;
}

static size_t staging_bytes_for(int world, int H, int W, int C) {
    if (H % world == 0) return 0;
    return (size_t)world * (size_t)(H / world + 1) * (size_t)W * (size_t)C * sizeof(float);
}

}  // namespace minerf

using namespace minerf;

extern "C" {

int mi_nerf_rccl_available(void) {
    const Rccl* r = nullptr;
    return rccl(&r);
}

int mi_nerf_comm_unique_id(void* id_host) {
    MN_CHECK_ARG(id_host != nullptr, "id_host is NULL");
    const Rccl* r = nullptr;
    if (int rc = rccl(&r)) return rc;
    RcclId id;
    if (int rc = r->GetUniqueId(&id)) return rccl_fail(r, rc, "ncclGetUniqueId");
    memcpy(id_host, &id, sizeof(id));
    return MI_NERF_OK;
}

int mi_nerf_comm_init_rank(const void* id_host, int world, int rank, void** comm_out) {
    MN_CHECK_ARG(comm_out != nullptr, "comm_out is NULL");
    *comm_out = nullptr;
    MN_CHECK_ARG(id_host != nullptr, "id_host is NULL");
    MN_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "rank %d of world %d", rank, world);
    const Rccl* r = nullptr;
    if (int rc = rccl(&r)) return rc;
    int dev = 0;
    MN_HIP(hipGetDevice(&dev));
    RcclId id;
    memcpy(&id, id_host, sizeof(id));
    RcclComm c = nullptr;
    if (int rc = r->CommInitRank(&c, world, id, rank)) return rccl_fail(r, rc, "ncclCommInitRank");
    TileComm* tc = new TileComm{c, world, rank, dev};
    {
        std::lock_guard<std::mutex> g(g_comms_mu);
        g_comms.insert(tc);
    }
    *comm_out = tc;
    return MI_NERF_OK;
}

static int check_comm(void* comm, TileComm** out) {
    MN_CHECK_ARG(comm != nullptr, "comm is NULL");
    TileComm* tc = (TileComm*)comm;
    {
        std::lock_guard<std::mutex> g(g_comms_mu);
        MN_CHECK_ARG(g_comms.count(tc) == 1, "comm is not a handle of mi_nerf_comm_init_rank (or was destroyed)");
    }
    *out = tc;
    return MI_NERF_OK;
}

int mi_nerf_comm_info(void* comm, int* world_out, int* rank_out, int* device_out) {
    TileComm* tc = nullptr;
    if (int rc = check_comm(comm, &tc)) return rc;
    if (world_out) *world_out = tc->world;
    if (rank_out) *rank_out = tc->rank;
    if (device_out) *device_out = tc->device;
    return MI_NERF_OK;
}

int mi_nerf_comm_destroy(void* comm) {
    MN_CHECK_ARG(comm != nullptr, "comm is NULL");
    TileComm* tc = (TileComm*)comm;
    {   // leave the registry first: a second destroy of the same handle (from any thread) is refused, never a double free
        std::lock_guard<std::mutex> g(g_comms_mu);
        MN_CHECK_ARG(g_comms.erase(tc) == 1, "comm is not a handle of mi_nerf_comm_init_rank (or was destroyed)");
    }
    const Rccl* r = nullptr;
    if (int rc = rccl(&r)) { delete tc; return rc; }
    const int rc = r->CommDestroy(tc->comm);
    delete tc;
    if (rc) return rccl_fail(r, rc, "ncclCommDestroy");
    return MI_NERF_OK;
}

size_t mi_nerf_all_gather_staging_bytes(int world, int H, int W, int C) {
    if (check_geometry(world, H, W, C)) return 0;
    return staging_bytes_for(world, H, W, C);
}

int mi_nerf_unpad_tiles(const float* staging_dev, int world, int H, int W, int C, float* frame_dev, void* stream) {
    return unpad_tiles(staging_dev, world, H, W, C, frame_dev, (hipStream_t)stream);
}

int mi_nerf_all_gather_tiles(void* comm, const float* tile_dev, int rows_local, int H, int W, int C, float* frame_dev, void* staging_dev,
                             size_t staging_bytes, void* stream) {
    TileComm* tc = nullptr;
    if (int rc = check_comm(comm, &tc)) return rc;
    if (int rc = check_geometry(tc->world, H, W, C)) return rc;
    MN_CHECK_ARG(tile_dev && frame_dev, "tile / frame is NULL");
    const int want = block_rows(H, tc->world, tc->rank);
    MN_CHECK_ARG(rows_local == want, "rank %d of %d owns %d of %d rows, got a tile of %d", tc->rank, tc->world, want, H, rows_local);
    int dev = 0;
    MN_HIP(hipGetDevice(&dev));
    MN_CHECK_ARG(dev == tc->device, "the communicator was created on device %d, the current device is %d", tc->device, dev);
    const Rccl* r = nullptr;
    if (int rc = rccl(&r)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const size_t row_floats = (size_t)W * C;
    {   // the tile either lies IN the frame at its own block (RCCL's in-place form) or does not touch it: anything between is undefined in RCCL
        const uintptr_t t0 = (uintptr_t)tile_dev, t1 = t0 + (size_t)rows_local * row_floats * sizeof(float);
        const uintptr_t f0 = (uintptr_t)frame_dev, f1 = f0 + (size_t)H * row_floats * sizeof(float);
        const uintptr_t own = f0 + (size_t)block_row0(H, tc->world, tc->rank) * row_floats * sizeof(float);
        MN_CHECK_ARG(t1 <= f0 || t0 >= f1 || t0 == own, "the tile overlaps the frame but is not this rank's block of it (in place means tile == frame + row0 * W * C)");
    }
    if (H % tc->world == 0) {                      // equal blocks: gather straight into the frame (in place when the tile already lies in it)
        if (int rc = r->AllGather(tile_dev, frame_dev, (size_t)rows_local * row_floats, 7 /* ncclFloat32 */, tc->comm, s))
            return rccl_fail(r, rc, "ncclAllGather");
        return MI_NERF_OK;
    }
    // ragged split: blocks padded to the largest one inside `staging`, gathered IN PLACE there, un-padded by one copy kernel
    const size_t need = staging_bytes_for(tc->world, H, W, C);
    MN_CHECK_ARG(staging_dev != nullptr && staging_bytes >= need, "staging: %zu bytes given, %zu needed (mi_nerf_all_gather_staging_bytes)", staging_bytes, need);
    MN_CHECK_ARG((uintptr_t)staging_dev % 16 == 0, "staging must be 16-byte aligned");
    {   // staging is written by the gather and read by the un-pad kernel while tile is read and frame written: it shares no byte with either
        const uintptr_t s0 = (uintptr_t)staging_dev, s1 = s0 + need;
        const uintptr_t t0 = (uintptr_t)tile_dev, t1 = t0 + (size_t)rows_local * row_floats * sizeof(float);
        const uintptr_t f0 = (uintptr_t)frame_dev, f1 = f0 + (size_t)H * row_floats * sizeof(float);
        MN_CHECK_ARG(t1 <= s0 || t0 >= s1, "the tile overlaps the staging buffer");
        MN_CHECK_ARG(f1 <= s0 || f0 >= s1, "the frame overlaps the staging buffer");
    }
    const size_t max_cnt = (size_t)(H / tc->world + 1) * row_floats;
    float* mine = (float*)staging_dev + (size_t)tc->rank * max_cnt;
    MN_HIP(hipMemcpyAsync(mine, tile_dev, (size_t)rows_local * row_floats * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (int rc = r->AllGather(mine, staging_dev, max_cnt, 7, tc->comm, s)) return rccl_fail(r, rc, "ncclAllGather");
    return unpad_tiles((const float*)staging_dev, tc->world, H, W, C, frame_dev, s);
}

}  // extern "C"
