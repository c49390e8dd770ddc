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
#include <string>
#include <vector>

#include "dvpari.h"

int main(int argc, char** argv) {
  std::vector<int> devices;
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
    } else {
      pos.push_back(argv[i]);
    }
  }
  if (pos.size() < 2) {
    fprintf(stderr, "usage: %s <cache_dir> <n_public> [<proof_out>] [--devices 0,1,..]\n", argv[0]);
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
