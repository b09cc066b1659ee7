// tests/hostemu/fake_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl.so.1 (that is its SONAME) for the host-emulation build, so that
// csrc/gpv_group.cpp's RCCL branch -- ncclCommInitAll / ncclCommInitRank, the in-place ncclAllGather of the packed accept bits with its status
// trailer, ncclCommCount / ncclCommUserRank -- runs with MORE THAN ONE RANK on a machine without GPUs. It moves bytes between "devices" that are
// host memory; it says nothing about RCCL itself (ncclGetVersion answers 0 so that no record can mistake it for the real library).
//
//   one process (ncclCommInitAll)     the ranks are threads of the process: the all-gather is a rendezvous over a shared table of pointers
//   one process per rank (InitRank)   the ranks meet in a POSIX shared-memory segment named after the unique id: every rank copies its slot in,
//                                     a process-shared barrier, every rank copies all slots out (slots <= 64 KiB, 64 ranks)
// A test preloads it (ctypes.CDLL(..., RTLD_GLOBAL)) before the first group call; gpv_group.cpp's dlopen("librccl.so.1", RTLD_NOLOAD) then binds it.
#include <rccl/rccl.h>

#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <condition_variable>
#include <mutex>
#include <vector>

namespace {
const size_t SHM_SLOT = 64 * 1024;
const int SHM_RANKS = 64;
struct Shm {  // one per unique id
  pthread_barrier_t barrier;
  int world;
  unsigned char slots[SHM_RANKS][SHM_SLOT];
};
struct Clique {  // in-process
  int n = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  unsigned long long generation = 0;
  std::vector<const void*> send;
  std::vector<void*> recv;
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const unsigned long long g = generation;
    if (++arrived == n) {
      arrived = 0;
      generation++;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != g; });
    }
  }
};
size_t type_bytes(ncclDataType_t t) { return t == ncclUint8 || t == ncclInt8 ? 1 : t == ncclInt32 || t == ncclUint32 ? 4 : 8; }
}  // namespace

struct ncclComm {
  int rank = 0, world = 1;
  Clique* clique = nullptr;  // in-process clique (shared by its n communicators; freed by the last one destroyed)
  int* clique_refs = nullptr;
  Shm* shm = nullptr;  // one process per rank
  char shm_name[80] = {0};
};

#define API extern "C" __attribute__((visibility("default")))

API ncclResult_t ncclGetVersion(int* v) { *v = 0; return ncclSuccess; }  // 0 = "not RCCL"
API const char* ncclGetErrorString(ncclResult_t e) { return e == ncclSuccess ? "no error" : "fake RCCL (tests/hostemu): error"; }
API ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof *id);
  timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  snprintf(id->internal, sizeof id->internal, "/gpv_hostemu_%d_%lld_%ld", (int)getpid(), (long long)ts.tv_sec, ts.tv_nsec);
  return ncclSuccess;
}
API ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int*) {
  if (n < 1) return ncclInvalidArgument;
  Clique* c = new Clique();
  c->n = n;
  c->send.assign((size_t)n, nullptr);
  c->recv.assign((size_t)n, nullptr);
  int* refs = new int(n);
  for (int r = 0; r < n; r++) {
    comms[r] = new ncclComm();
    comms[r]->rank = r;
    comms[r]->world = n;
    comms[r]->clique = c;
    comms[r]->clique_refs = refs;
  }
  return ncclSuccess;
}
API ncclResult_t ncclCommInitRank(ncclComm_t* comm, int world, ncclUniqueId id, int rank) {
  if (world < 1 || world > SHM_RANKS || rank < 0 || rank >= world) return ncclInvalidArgument;
  ncclComm* c = new ncclComm();
  c->rank = rank;
  c->world = world;
  if (world > 1) {
    id.internal[sizeof id.internal - 1] = 0;
    snprintf(c->shm_name, sizeof c->shm_name, "%s", id.internal);
    // rank 0 creates and initialises the segment; the others wait for it to reach its size (the barrier is initialised before the size is set)
    int fd = -1;
    if (rank == 0) {
      fd = shm_open(c->shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0) { delete c; return ncclSystemError; }
      // build it under a private mapping of a temporary size, publish by growing to the final size last
      if (ftruncate(fd, (off_t)sizeof(Shm) + 1) != 0) { close(fd); delete c; return ncclSystemError; }
    } else {
      for (int tries = 0; tries < 30000 && fd < 0; tries++) {
        fd = shm_open(c->shm_name, O_RDWR, 0600);
        if (fd < 0) usleep(1000);
      }
      if (fd < 0) { delete c; return ncclSystemError; }
      struct stat sb;  // ... and for rank 0 to have given it its size
      for (int tries = 0; tries < 30000; tries++) {
        if (fstat(fd, &sb) == 0 && (size_t)sb.st_size >= sizeof(Shm) + 1) break;
        usleep(1000);
      }
      if ((size_t)sb.st_size < sizeof(Shm) + 1) { close(fd); delete c; return ncclSystemError; }
    }
    void* p = mmap(nullptr, sizeof(Shm) + 1, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (p == MAP_FAILED) { close(fd); delete c; return ncclSystemError; }
    c->shm = (Shm*)p;
    volatile unsigned char* ready = (volatile unsigned char*)p + sizeof(Shm);
    if (rank == 0) {
      pthread_barrierattr_t a;
      pthread_barrierattr_init(&a);
      pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
      pthread_barrier_init(&c->shm->barrier, &a, (unsigned)world);
      c->shm->world = world;
      __atomic_store_n(ready, (unsigned char)1, __ATOMIC_RELEASE);
    } else {
      for (int tries = 0; tries < 30000 && !__atomic_load_n(ready, __ATOMIC_ACQUIRE); tries++) usleep(1000);
      if (!*ready || c->shm->world != world) { munmap(p, sizeof(Shm) + 1); close(fd); delete c; return ncclSystemError; }
    }
    close(fd);
    pthread_barrier_wait(&c->shm->barrier);  // everybody is attached: the name can go
    if (rank == 0) shm_unlink(c->shm_name);
  }
  *comm = c;
  return ncclSuccess;
}
API ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclInvalidArgument;
  if (c->clique && --*c->clique_refs == 0) { delete c->clique; delete c->clique_refs; }
  if (c->shm) munmap(c->shm, sizeof(Shm) + 1);
  delete c;
  return ncclSuccess;
}
API ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { *n = c->world; return ncclSuccess; }
API ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r) { *r = c->rank; return ncclSuccess; }
API ncclResult_t ncclGroupStart() { return ncclSuccess; }
API ncclResult_t ncclGroupEnd() { return ncclSuccess; }
// recv[r * count .. (r + 1) * count) = rank r's send, on every rank; send may be recv + rank * count (in place)
API ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t c, hipStream_t) {
  const size_t bytes = count * type_bytes(type);
  if (c->world == 1) {
    if ((const char*)send != (char*)recv) memmove(recv, send, bytes);
    return ncclSuccess;
  }
  if (c->clique) {
    Clique* q = c->clique;
    q->send[(size_t)c->rank] = send;
    q->recv[(size_t)c->rank] = recv;
    q->barrier();  // every rank's pointers are published and its slot is complete
    for (int r = 0; r < c->world; r++)
      if (r != c->rank) memcpy((char*)recv + (size_t)r * bytes, q->send[(size_t)r], bytes);
    if ((const char*)send != (char*)recv + (size_t)c->rank * bytes) memmove((char*)recv + (size_t)c->rank * bytes, send, bytes);
    q->barrier();  // nobody's send buffer is reused before everyone has read it
    return ncclSuccess;
  }
  if (!c->shm || bytes > SHM_SLOT) return ncclInvalidUsage;
  memcpy(c->shm->slots[c->rank], send, bytes);
  pthread_barrier_wait(&c->shm->barrier);
  for (int r = 0; r < c->world; r++) memcpy((char*)recv + (size_t)r * bytes, c->shm->slots[r], bytes);
  pthread_barrier_wait(&c->shm->barrier);
  return ncclSuccess;
}
