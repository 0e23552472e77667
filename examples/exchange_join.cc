// Multi-GPU use of the executor from C++: one process per GPU, the partition exchange
// inside libhmj_hip.so.  Host-only C++11 (g++, no hipcc, no MPI): the launcher passes rank, world size and the
// path of a file through which rank 0 hands its 128-byte communicator id to the others.
//
//   for r in 0 1; do ./examples/exchange_join $r 2 /tmp/hmj.id 26 & done; wait
//
// Every rank generates its row shard of the benchmark relations on its GPU (the same "same key set, two orders"
// property as the reference's r = create_strvec(n), s = create_strvec(n), hashjoin_bench.cc:112-113), joins, and
// prints its share and the reduction over all ranks -- sum(rval + sval), what hashjoin_bench.cc:131-133 computes.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <thread>

#include "hmj.h"

extern "C" {  // the HIP runtime: device memory for the shards (the library itself never allocates the caller's inputs)
int hipSetDevice(int);
int hipMalloc(void**, size_t);
int hipFree(void*);
}

static int die(hmj_ctx* c, const char* what, int rc) {
  std::fprintf(stderr, "%s failed: %s (%s)\n", what, hmj_strerror(rc), c ? hmj_last_error(c) : "");
  return 1;
}

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s rank world id_file [log2 rows per rank = 24] [flags = 0]\n", argv[0]);
    return 2;
  }
  const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
  const char* id_file = argv[3];
  const uint64_t n = 1ull << (argc > 4 ? std::atoi(argv[4]) : 24);
  const uint32_t flags = argc > 5 ? (uint32_t)std::atoi(argv[5]) : 0;
  hmj_ctx* ctx = nullptr;
  int rc = hmj_create(&ctx, rank);  // GPU `rank` of this node
  if (rc) return die(nullptr, "hmj_create", rc);

  char id[HMJ_UNIQUE_ID_BYTES];
  if (rank == 0) {
    if ((rc = hmj_comm_unique_id(id))) return die(ctx, "hmj_comm_unique_id", rc);
    std::ofstream f(std::string(id_file) + ".tmp", std::ios::binary);
    f.write(id, sizeof id);
    f.close();
    std::rename((std::string(id_file) + ".tmp").c_str(), id_file);  // appears complete or not at all
  } else {
    for (int tries = 0;; tries++) {
      std::ifstream f(id_file, std::ios::binary);
      if (f && f.read(id, sizeof id)) break;
      if (tries > 600) return die(ctx, "waiting for the communicator id", HMJ_E_RCCL);
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
  }
  if ((rc = hmj_comm_init_rank(ctx, world, rank, id))) return die(ctx, "hmj_comm_init_rank", rc);

  void *r_shard = nullptr, *s_shard = nullptr;
  hipSetDevice(rank);
  if (hipMalloc(&r_shard, n * 16) || hipMalloc(&s_shard, n * 16)) return die(ctx, "hipMalloc", HMJ_E_OOM);
  const uint64_t seed = 0x243F6A8885A308D3ull, n_total = n * (uint64_t)world;
  if ((rc = hmj_gen_build_u64_device(ctx, r_shard, n, (uint64_t)rank * n, seed))) return die(ctx, "gen build", rc);
  if ((rc = hmj_gen_probe_u64_device(ctx, s_shard, n, (uint64_t)rank * n, n_total, seed, 0))) return die(ctx, "gen probe", rc);

  hmj_result mine, all;
  for (int it = 0; it < 3; it++) {
    const auto t0 = std::chrono::steady_clock::now();
    if ((rc = hmj_exchange_join_u64_device(ctx, r_shard, n, s_shard, n, flags, &mine, &all))) return die(ctx, "hmj_exchange_join_u64_device", rc);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    hmj_exchange_info info;
    hmj_last_exchange_info(ctx, &info);
    std::printf("rank %d/%d: owns %llu build + %llu probe rows, %llu matches; all ranks: %llu matches, sum %llu; %.2f ms "
                "(split %.2f, exchange %.2f + %.2f, local %.2f)\n", rank, world, (unsigned long long)info.recv_build,
                (unsigned long long)info.recv_probe, (unsigned long long)mine.n_matches, (unsigned long long)all.n_matches,
                (unsigned long long)(all.sum_r + all.sum_s), ms, info.ms_split, info.ms_exchange_build, info.ms_exchange_probe,
                info.ms_local);
  }
  const bool ok = all.n_matches == n_total;  // every probe row matches exactly one build row
  hipFree(r_shard);
  hipFree(s_shard);
  hmj_destroy(ctx);
  return ok ? 0 : 1;
}
