// Host program over the C ABI only (include/dvpari.h): what a compiled-language maintainer -- the reference is Rust -- links
// against.  It mirrors the reference's `Proof::prove(cache_dir, public_inputs, private_inputs)` call
// (src/proving.rs:426) from files alone:
//
//   dvp_prove_cli <cache_dir> <n_public> [<proof_out>] [--devices 0,1,2,3]
//
// --devices d0,d1,..: in-library multi-GPU (dvp_set_devices): the two MSMs of the proof are sharded over the listed
// devices, one host thread each; d0 is the home device.  An id may repeat.  The proof bytes do not depend on the list.
//
// reads <cache_dir>/witness_to_dvsnark (u32-BE count || 32-byte BE elements = [1, public.., private..],
// src/gnark_r1cs.rs:188-210), lets the library open the R1CS dump and the SRS point files of the same directory, proves on
// the GPU and prints the 118 proof bytes in hex (optionally also writes them to <proof_out>).
//
// build:  g++ -O2 -std=c++17 -Iinclude examples/dvp_prove_cli.cpp -Ldv-pari_amd -ldvpari_hip -Wl,-rpath,$PWD/dv-pari_amd -o dvp_prove_cli
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "dvpari.h"

int main(int argc, char** argv) {
  std::vector<int> devices;
  int repeat = 0, threads = 1;  // --repeat N --threads T: N more proofs through dvp_prove_cache_dir from T host threads
  std::vector<char*> pos;
  for (int i = 1; i < argc; ++i) {
    if (std::string(argv[i]) == "--devices" && i + 1 < argc) {
      for (char* tok = argv[++i]; *tok;) {
        char* end = tok;
        const long id = strtol(tok, &end, 10);
        if (end == tok || (*end != ',' && *end != 0) || id < 0 || id > 1023) {  // not a number / stray character: do not spin on it
          fprintf(stderr, "--devices: expected a comma-separated list of device ids, got '%s'\n", argv[i]);
          return 2;
        }
        devices.push_back((int)id);
        tok = *end == ',' ? end + 1 : end;
      }
    } else if ((std::string(argv[i]) == "--repeat" || std::string(argv[i]) == "--threads") && i + 1 < argc) {
      const bool rep = std::string(argv[i]) == "--repeat";
      char* end = nullptr;
      const long v = strtol(argv[++i], &end, 10);
      if (end == argv[i] || *end != 0 || v < 1 || v > 1000000) {
        fprintf(stderr, "%s: expected a positive number, got '%s'\n", rep ? "--repeat" : "--threads", argv[i]);
        return 2;
      }
      (rep ? repeat : threads) = (int)v;
    } else {
      pos.push_back(argv[i]);
    }
  }
  if (pos.size() < 2) {
    fprintf(stderr, "usage: %s <cache_dir> <n_public> [<proof_out>] [--devices 0,1,..] [--repeat N --threads T]\n", argv[0]);
    return 2;
  }
  argc = (int)pos.size() + 1;
  for (size_t i = 0; i < pos.size(); ++i) argv[i + 1] = pos[i];
  const std::string dir = argv[1];
  const uint32_t n_public = (uint32_t)strtoul(argv[2], nullptr, 10);
  if (dvp_device_count() <= 0) {
    fprintf(stderr, "no HIP device visible (the library has no CPU path)\n");
    return 3;
  }
  if (!devices.empty()) {
    int rcd = dvp_set_device(devices[0]);
    if (rcd == DVP_OK) rcd = dvp_set_devices(devices.data(), (int)devices.size());
    if (rcd != DVP_OK) {
      fprintf(stderr, "--devices: %s\n", dvp_strerror(rcd));
      return 1;
    }
  }
  const std::string wpath = dir + "/witness_to_dvsnark";
  size_t n = 0;
  int rc = dvp_file_witness_read(wpath.c_str(), nullptr, 0, &n);
  if (rc != DVP_OK) {
    fprintf(stderr, "%s: %s\n", wpath.c_str(), dvp_strerror(rc));
    return 1;
  }
  std::vector<uint64_t> w(4 * n);
  rc = dvp_file_witness_read(wpath.c_str(), w.data(), n, &n);
  if (rc != DVP_OK || n < 1 + (size_t)n_public) {
    fprintf(stderr, "witness: %s (n = %zu)\n", dvp_strerror(rc), n);
    return 1;
  }
  if (w[0] != 1 || w[1] || w[2] || w[3]) {
    fprintf(stderr, "witness[0] must be the constant 1 (src/proving.rs:449-452)\n");
    return 1;
  }
  uint8_t proof[118];
  // The dump knows wires only up to the highest one used; the library takes the witness length from |g_m|, so hand over
  // exactly that many private inputs (a longer witness file is truncated by the reference's zip as well).
  dvp_prover* p = nullptr;
  rc = dvp_prover_open_cache_dir(dir.c_str(), n_public, &p);
  if (rc != DVP_OK) {
    fprintf(stderr, "open %s: %s (index %lld)\n", dir.c_str(), dvp_strerror(rc), (long long)dvp_last_error_index());
    return 1;
  }
  // |[w | q2]| = n_wires + m and |[k_a | k_b | k_r]| = 4m
  const size_t n_wires = dvp_prover_msm_size(p, 0) - dvp_prover_msm_size(p, 1) / 4;
  if (n_wires < 1 + (size_t)n_public || n_wires > n) {
    fprintf(stderr, "witness has %zu entries, the commitment key wants %zu\n", n, n_wires);
    dvp_prover_destroy(p);
    return 1;
  }
  rc = dvp_prove(p, w.data() + 4, n_public, w.data() + 4 * (1 + (size_t)n_public), (uint32_t)(n_wires - 1 - n_public), proof);
  dvp_prover_destroy(p);
  if (rc != DVP_OK) {
    fprintf(stderr, "prove: %s (index %lld)\n", dvp_strerror(rc), (long long)dvp_last_error_index());
    return 1;
  }
  for (int i = 0; i < 118; ++i) printf("%02x", proof[i]);
  printf("\n");
  if (repeat > 0) {
    // Throughput through the reference's own signature (Proof::prove(cache_dir, public, private) = dvp_prove_cache_dir): `threads`
    // host threads share `repeat` proofs; with two or more, the library keeps two proofs in flight on the GPU.  Every proof must
    // equal the first one.
    const uint64_t* pub = w.data() + 4;
    const uint64_t* prv = w.data() + 4 * (1 + (size_t)n_public);
    const uint32_t n_prv = (uint32_t)(n_wires - 1 - n_public);
    uint8_t warm[118];
    for (int t = 0; t < 2; ++t) {  // opens the cache_dir entry (and, below, its second prover) outside the timed part
      rc = dvp_prove_cache_dir(dir.c_str(), pub, n_public, prv, n_prv, warm);
      if (rc != DVP_OK || memcmp(warm, proof, 118)) {
        fprintf(stderr, "dvp_prove_cache_dir: %s\n", rc ? dvp_strerror(rc) : "bytes differ from dvp_prove");
        return 1;
      }
    }
    std::atomic<int> next(0), bad(0);
    auto worker = [&](int limit) {
      uint8_t out[118];
      while (next.fetch_add(1) < limit) {
        const int r = dvp_prove_cache_dir(dir.c_str(), pub, n_public, prv, n_prv, out);
        if (r != DVP_OK || memcmp(out, proof, 118)) bad.fetch_add(1);
      }
    };
    auto run = [&](int limit) {
      next = 0;
      std::vector<std::thread> th;
      for (int t = 0; t < threads; ++t) th.emplace_back(worker, limit);
      for (auto& t : th) t.join();
    };
    run(2 * threads);  // concurrent warm-up: the second prover of the entry is opened when two calls first overlap
    const auto t0 = std::chrono::steady_clock::now();
    run(repeat);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (bad.load()) {
      fprintf(stderr, "%d of the repeated proofs failed or differ\n", bad.load());
      return 1;
    }
    fprintf(stderr, "%d proofs from %d host thread(s): %.2f ms per proof (witness in host memory, dvp_prove_cache_dir)\n", repeat, threads, ms / repeat);
    dvp_cache_dir_release(nullptr);
  }
  if (argc > 3) {
    FILE* f = fopen(argv[3], "wb");
    if (!f || fwrite(proof, 1, 118, f) != 118) {
      fprintf(stderr, "cannot write %s\n", argv[3]);
      return 1;
    }
    fclose(f);
  }
  return 0;
}
