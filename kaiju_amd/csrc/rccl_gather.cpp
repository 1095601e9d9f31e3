// rccl_gather.cpp — the one collective of the path, in the product: per-GPU 16-byte records gathered on one rank with a single
// RCCL gather over xGMI (BASELINE north star; the reference has no exchange at all: its threads write to one stream under a
// mutex, ConsumerThread.cpp:847-856, kaiju.cpp:250-257).  One process per GPU; the processes of a node find each other through
// a file (rank 0 leaves the communicator's id there).  librccl is opened at the first use (dlopen): the library and the
// command line programs do not link it, a single-GPU run never loads it.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <algorithm>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/kaiju_gpu.h"

namespace {

// the part of rccl.h this file needs (the header itself pulls in HIP device headers)
constexpr int kUniqueIdBytes = 128;                 // NCCL_UNIQUE_ID_BYTES
struct UniqueId { char internal[kUniqueIdBytes]; };
typedef void *Comm;
constexpr int kNcclSuccess = 0, kNcclUint8 = 1;
struct Rccl {
  void *lib = nullptr;
  int (*GetUniqueId)(UniqueId *) = nullptr;
  int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*Gather)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;      // RCCL's own (rccl.h: ncclGather)
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, Comm, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string err, path;
  bool load() {
    if (lib) return true;
    // The RCCL that belongs to the HIP runtime THIS process runs on: a process may hold two ROCm stacks (a Python host: the
    // system's under /opt/rocm and the one bundled with torch), whichever libamdhip64 was loaded first serves everybody - and
    // an RCCL of the other stack would bring up a second HSA runtime that finds no device ("no ROCm-capable device is
    // detected" from ncclCommInitRank).  So: first the librccl that lies next to the libamdhip64 in use, then the usual names.
    std::string beside;
    {
      Dl_info info;
      if (dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
        beside = info.dli_fname;
        const size_t slash = beside.rfind('/');
        beside = slash == std::string::npos ? std::string() : beside.substr(0, slash + 1);
      }
    }
    const std::string cands[] = {beside.empty() ? std::string() : beside + "librccl.so.1", beside.empty() ? std::string() : beside + "librccl.so",
                                 "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string tried;
    for (const std::string &name : cands) {
      if (name.empty()) continue;
      lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (lib) { path = name; break; }
      const char *e = dlerror();                      // (ONE call: glibc clears the message when it is read)
      tried += std::string(tried.empty() ? "" : "; ") + (e ? e : name.c_str());
    }
    if (!lib) { err = std::string("librccl not found: ") + tried; return false; }
    auto sym = [&](const char *n) { return dlsym(lib, n); };
    GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
    CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    Gather = reinterpret_cast<decltype(Gather)>(sym("ncclGather"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
    Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    if (!GetUniqueId || !CommInitRank || !CommDestroy || (!Gather && !(GroupStart && GroupEnd && Send && Recv))) {
      err = "librccl lacks the entry points of a gather"; dlclose(lib); lib = nullptr; return false;
    }
    return true;
  }
};
Rccl g_rccl;
std::mutex g_mu;

thread_local std::string tl_err;
int fail(int code, const std::string &m) { tl_err = m; return code; }
double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

// rank 0 hands `*id` to the other ranks of THIS job through files (see below); KAIJU_GPU_OK or an error code + tl_err
int exchange_id(const char *rendezvous_path, int rank, int world, UniqueId *id, double timeout_s) {
  // The communicator's id travels through files.  A path may have been used before (a job that died, a rank that polls before
  // rank 0 has started): nothing that lies there is trusted.  Rank r > 0 leaves a random 64-bit nonce in `<path>.r<r>`; rank 0
  // removes whatever `<path>` held, makes the id and writes {magic, world, the nonces it has read, id} (next to it, then
  // renamed: a reader never sees half of it); rank r takes the id only from a file that carries ITS nonce and then removes its
  // nonce file - the acknowledgement rank 0 waits for (re-reading the nonces meanwhile: one left by a dead job is replaced by
  // its rank and the file is written again).  When every rank has acknowledged, rank 0 removes the file.
  const std::string base = rendezvous_path;
  auto nonce_path = [&](int r) { return base + ".r" + std::to_string(r); };
  auto write_atomically = [&](const std::string &path, const void *data, size_t bytes) {
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    FILE *fp = fopen(tmp.c_str(), "wb");
    const bool ok = fp && fwrite(data, bytes, 1, fp) == 1;
    if (fp) fclose(fp);
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) { unlink(tmp.c_str()); return false; }
    return true;
  };
  auto read_whole = [&](const std::string &path, void *data, size_t bytes) {
    struct stat st;
    if (stat(path.c_str(), &st) != 0 || st.st_size != (off_t)bytes) return false;
    FILE *fp = fopen(path.c_str(), "rb");
    const bool ok = fp && fread(data, bytes, 1, fp) == 1;
    if (fp) fclose(fp);
    return ok;
  };
  constexpr uint64_t kMagic = 0x4b4a434f4d4d3031ull;      // "KJCOMM01"
  const size_t file_words = 2 + (size_t)world;             // magic, world, nonce of rank 1 .. world-1 (slot 0 unused)
  const size_t file_bytes = file_words * 8 + sizeof *id;
  std::vector<uint8_t> buf(file_bytes, 0);
  const double t0 = now_s();
  const double kWait = timeout_s;
  if (rank == 0) {
    std::vector<uint64_t> nonce(world, 0), seen(world, 0);
    std::vector<char> acked(world, 0);
    bool written = false;
    for (;;) {
      bool all_known = true, all_acked = true, changed = false;
      for (int r = 1; r < world; r++) {
        uint64_t v = 0;
        if (read_whole(nonce_path(r), &v, sizeof v)) {
          if (!seen[r] || v != nonce[r]) { nonce[r] = v; seen[r] = 1; changed = true; }
          all_acked = false;
        } else if (written && seen[r]) acked[r] = 1;         // its nonce file is gone: the rank has the id
        if (!seen[r]) all_known = false;
        if (!acked[r]) all_acked = false;
      }
      if (all_known && (changed || !written)) {
        uint64_t *w = reinterpret_cast<uint64_t *>(buf.data());
        w[0] = kMagic; w[1] = (uint64_t)world;
        for (int r = 1; r < world; r++) w[2 + r] = nonce[r];
        memcpy(buf.data() + file_words * 8, id, sizeof *id);
        if (!write_atomically(base, buf.data(), file_bytes)) return fail(KAIJU_GPU_ERR_IO, std::string("cannot write ") + rendezvous_path);
        written = true;
        std::fill(acked.begin(), acked.end(), 0);
        continue;
      }
      if (written && all_acked) break;
      if (now_s() - t0 > kWait) { unlink(rendezvous_path); return fail(KAIJU_GPU_ERR_IO, std::string("not every rank answered through ") + rendezvous_path); }
      usleep(2000);
    }
    unlink(rendezvous_path);                                 // every rank has acknowledged: nothing is left behind
  } else {
    uint64_t mine = 0;
    {
      FILE *fp = fopen("/dev/urandom", "rb");
      if (!fp || fread(&mine, sizeof mine, 1, fp) != 1) mine = 0;
      if (fp) fclose(fp);
      timespec t; clock_gettime(CLOCK_REALTIME, &t);
      if (!mine) mine = ((uint64_t)t.tv_nsec << 32) ^ (uint64_t)t.tv_sec ^ ((uint64_t)getpid() << 17);
      mine |= 1;                                             // (never 0)
    }
    if (!write_atomically(nonce_path(rank), &mine, sizeof mine)) return fail(KAIJU_GPU_ERR_IO, std::string("cannot write ") + nonce_path(rank));
    for (;;) {
      if (read_whole(base, buf.data(), file_bytes)) {
        const uint64_t *w = reinterpret_cast<const uint64_t *>(buf.data());
        if (w[0] == kMagic && w[1] == (uint64_t)world && w[2 + rank] == mine) { memcpy(id, buf.data() + file_words * 8, sizeof *id); break; }
      }
      if (now_s() - t0 > kWait) { unlink(nonce_path(rank).c_str()); return fail(KAIJU_GPU_ERR_IO, std::string("no communicator id for this job appeared in ") + rendezvous_path); }
      usleep(2000);
    }
    unlink(nonce_path(rank).c_str());
  }
  return KAIJU_GPU_OK;
}

}  // namespace

struct kaiju_gpu_comm {
  Comm comm = nullptr;
  int rank = 0, world = 1, device = 0;
};

extern "C" const char *kaiju_gpu_comm_last_error(void) { return tl_err.c_str(); }

extern "C" int kaiju_gpu_comm_create(const char *rendezvous_path, int rank, int world, int device_id, kaiju_gpu_comm **out) {
  if (!out) return fail(KAIJU_GPU_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (!rendezvous_path || world < 1 || rank < 0 || rank >= world) return fail(KAIJU_GPU_ERR_ARG, "bad rank / world / rendezvous path");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(KAIJU_GPU_ERR_NO_DEVICE, "hipGetDeviceCount found no device");
  if (device_id < 0 || device_id >= ndev) return fail(KAIJU_GPU_ERR_ARG, "device_id out of range");
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_rccl.load()) return fail(KAIJU_GPU_ERR_UNSUPPORTED, g_rccl.err);
  }
  if (hipSetDevice(device_id) != hipSuccess) return fail(KAIJU_GPU_ERR_HIP, "hipSetDevice");
  // (RCCL tests hipGetLastError() behind its own launches: an error code left behind by an earlier call of this process - a
  //  probing hipMalloc that was allowed to fail, an event query that said "not ready" - must not become its "unhandled error")
  (void)hipDeviceSynchronize();
  (void)hipGetLastError();
  UniqueId id;
  memset(&id, 0, sizeof id);
  if (rank == 0) {
    unlink(rendezvous_path);
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc != kNcclSuccess) return fail(KAIJU_GPU_ERR_HIP, std::string("ncclGetUniqueId: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"));
  }
  if (const int rc = exchange_id(rendezvous_path, rank, world, &id, 120.0)) return rc;
  kaiju_gpu_comm *c = new kaiju_gpu_comm();
  c->rank = rank; c->world = world; c->device = device_id;
  const int rc = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (rc != kNcclSuccess) { delete c; return fail(KAIJU_GPU_ERR_HIP, std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?")); }
  *out = c;
  return KAIJU_GPU_OK;
}

/* the file exchange alone (host only; tests/test_capi.py: stale files of an earlier job, ranks that start in any order):
   rank 0 passes 128 bytes in, the others get them */
extern "C" int kaiju_gpu_comm_exchange_id(const char *rendezvous_path, int rank, int world, uint8_t *id128, double timeout_s) {
  if (!rendezvous_path || !id128 || world < 1 || rank < 0 || rank >= world) return fail(KAIJU_GPU_ERR_ARG, "bad argument");
  UniqueId id;
  memcpy(&id, id128, sizeof id);
  if (rank == 0) unlink(rendezvous_path);
  const int rc = exchange_id(rendezvous_path, rank, world, &id, timeout_s);
  if (rc == KAIJU_GPU_OK) memcpy(id128, &id, sizeof id);
  return rc;
}

extern "C" void kaiju_gpu_comm_destroy(kaiju_gpu_comm *c) {
  if (!c) return;
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  delete c;
}

/* which librccl the gather runs on (diagnostics; "" before the first communicator) */
extern "C" const char *kaiju_gpu_comm_library(void) { return g_rccl.path.c_str(); }
extern "C" int kaiju_gpu_comm_rank(const kaiju_gpu_comm *c) { return c ? c->rank : -1; }
extern "C" int kaiju_gpu_comm_world(const kaiju_gpu_comm *c) { return c ? c->world : 0; }

// n records of 16 bytes from every rank into d_recv of `root` (rank r's at d_recv + r * n); asynchronous on `stream`
extern "C" int kaiju_gpu_gather_compact(kaiju_gpu_comm *c, const kaiju_gpu_compact *d_send, uint32_t n, kaiju_gpu_compact *d_recv,
                                        int root, void *stream) {
  if (!c || !c->comm || root < 0 || root >= c->world || (n && !d_send) || (c->rank == root && n && !d_recv))
    return fail(KAIJU_GPU_ERR_ARG, "bad argument");
  if (hipSetDevice(c->device) != hipSuccess) return fail(KAIJU_GPU_ERR_HIP, "hipSetDevice");
  hipStream_t s = static_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  const size_t bytes = (size_t)n * sizeof(kaiju_gpu_compact);
  int rc = kNcclSuccess;
  if (g_rccl.Gather) rc = g_rccl.Gather(d_send, d_recv, bytes, kNcclUint8, root, c->comm, s);     // ONE collective
  else {
    // the same as grouped point-to-point calls (what ncclGather is inside): still one group = one launch
    rc = g_rccl.GroupStart();
    if (rc == kNcclSuccess && c->rank == root)
      for (int r = 0; r < c->world && rc == kNcclSuccess; r++)
        rc = g_rccl.Recv(reinterpret_cast<uint8_t *>(d_recv) + (size_t)r * bytes, bytes, kNcclUint8, r, c->comm, s);
    if (rc == kNcclSuccess) rc = g_rccl.Send(d_send, bytes, kNcclUint8, root, c->comm, s);
    const int rc2 = g_rccl.GroupEnd();
    if (rc == kNcclSuccess) rc = rc2;
  }
  if (rc != kNcclSuccess) return fail(KAIJU_GPU_ERR_HIP, std::string("RCCL gather: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"));
  return KAIJU_GPU_OK;
}
