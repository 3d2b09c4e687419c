// A stand-in for librccl, for tests only: ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy / ncclGetErrorString between
// PROCESSES THAT SHARE ONE GPU, through a POSIX shared-memory segment on the host.  RCCL itself refuses two ranks on one device, and the test
// box has one GPU, so everything mi_nerf_all_gather_tiles does AROUND the collective for world sizes > 1 -- rank / world plumbing, the padded
// in-place gather inside the staging buffer, the un-pad kernel's block map, the in-place form on the frame -- would otherwise never run with
// more than one rank before the first real multi-GPU job.  Loaded through MI_NERF_RCCL_LIB (csrc/comm.hip resolves librccl at first use).
// Synchronous (the stream is drained first): this checks placement, not overlap.  tests/test_gpu_dist_nccl.py builds it with hipcc.
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

namespace {
constexpr size_t SLOT_BYTES = 32u << 20;          // per-rank exchange slot (an 800 x 800 x 4 fp32 frame is 10 MB)
struct Header { volatile int arrived; int world; };
struct Comm { int world, rank; long generation; char name[64]; char* base; size_t bytes; };
struct Id { char internal[128]; };

void barrier(Comm* c) {
    c->generation += 1;
    __sync_add_and_fetch(&((Header*)c->base)->arrived, 1);
    while (((Header*)c->base)->arrived < c->generation * c->world) usleep(50);
}
}  // namespace

extern "C" {

int ncclGetUniqueId(Id* id) {
    memset(id, 0, sizeof(*id));
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof(id->internal), "/minerf_fake_rccl_%d_%ld", (int)getpid(), (long)ts.tv_nsec);
    return 0;
}

int ncclCommInitRank(void** comm, int world, Id id, int rank) {
    Comm* c = new Comm();
    c->world = world; c->rank = rank; c->generation = 0;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    c->bytes = 4096 + (size_t)world * SLOT_BYTES;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) return 2;          // zero-filled by the kernel: arrived starts at 0 for whoever comes first
    c->base = (char*)mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->base == MAP_FAILED) return 2;
    barrier(c);
    *comm = c;
    return 0;
}

int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    static int calls = 0;                                                 // FAKE_RCCL_HANG_AFTER=n: the (n+1)-th gather of this process never returns
    const char* hang = getenv("FAKE_RCCL_HANG_AFTER");                    // (what a wedged collective looks like to the caller: bench.py's watchdog test)
    if (hang && ++calls > atoi(hang)) for (;;) sleep(1);
    const size_t bytes = count * 4;
    if (dtype != 7 || bytes > SLOT_BYTES) return 4;                      // ncclFloat32 only
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    char* slots = c->base + 4096;
    if (hipMemcpy(slots + (size_t)c->rank * SLOT_BYTES, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    barrier(c);                                                           // every rank's block is in its slot
    for (int r = 0; r < c->world; ++r)
        if (hipMemcpy((char*)recv + (size_t)r * bytes, slots + (size_t)r * SLOT_BYTES, bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
    barrier(c);                                                           // nobody refills a slot before everyone has read it
    return 0;
}

int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    barrier(c);
    munmap(c->base, c->bytes);
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
    return 0;
}

const char* ncclGetErrorString(int rc) { return rc == 0 ? "no error" : (rc == 1 ? "fake rccl: HIP call failed" : "fake rccl: error"); }

}  // extern "C"
