#!/usr/bin/env python3
"""
bench.py -- DV-Pari prover hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
      N > 1 with nothing around it: this process starts the N ranks itself (python -m torch.distributed.run, one rank
      per GPU over RCCL) BEFORE it touches a GPU, relays rank 0's JSON line and exits with the ranks' status.  Under a
      launcher (WORLD_SIZE set, e.g. the driver's torch.distributed.run) it is one of the ranks.
  python bench.py --gpus N --inproc          ONE process, in-library multi-GPU: dvp_set_devices(0..N-1)

One "step" = one full Proof::prove (src/proving.rs:426-688) of a synthetic dense R1CS with 2^20 constraints (BASELINE
config #4, the configuration the metric "R1CS constraints/sec (prove) at 2^20" is quoted on), witness already resident
in HBM.  The K steps run ONE PROOF AT A TIME and that is `value` / `ms_per_step` (= the latency of a proof): the quantity
the reference's prove() defines, and the loop the stage breakdown, the roofline block and the committed profiles describe.
After the timed loop, outside it, the same proof is also run with two provers in flight on the GPU (own host thread and
stream each, every proof byte-compared): `throughput_two_in_flight` -- a proving service's sustained rate, never `value`
(--in-flight 1 skips that leg).  With N > 1 the two MSMs of the proof are sharded by index range over the ranks and combined by all-gather +
local add (strong scaling: the proof size is fixed); the line then also carries what ran (`rccl_ranks`, `backend`, every
rank's own ms per step) and, beside it, the in-library figure for the same device count (`ms_per_step_inproc`,
dvp_set_devices: one process, one host thread per device, peer copies -- measured by a child process after the ranks
have released their GPUs).  Rank 0 prints ONE JSON line; the proof of the last step is checked with the
designated-verifier equation (src/srs.rs:374-428) outside the timed region.

roofline (dominant kernel = dvp::k_affine_round<true>, the first batched-affine pair round of each MSM: it gathers every
base once per window).  `achieved` / `peak` / `frac` are the contract's figure -- algorithmic bytes (96 B per (scalar,
base) pair, SURVEY 8d, x pairs per launch) / the launch time measured live with HIP events on the launch stream, against
the 8 TB/s HBM peak; it is tiny by construction.  What limits the kernel is stated by two models measured IN THIS RUN,
outside the timed loop, and `bound` names the larger fraction:
  * gather_model: 64-byte line gathers per launch (two operands per addition, read in both passes of the shared-inversion
    trick) / launch time, against dvp_ubench_gather = random 64-byte reads per second out of the prover's own table;
  * work_model: field-product equivalents per launch / launch time, against dvp_ubench_gf_mul = the multiplier alone.
`traffic` / `issue` come from the committed rocprofv3 PMC passes of this same command (profiles/): counters cannot be
collected inside an un-profiled run.
cpu_baseline: the C restatement with the reference's algorithmic shape (oracle/dvp_oracle.c: one tau-adic scalar
multiplication per point -- width-5 tau-NAF, PCLMULQDQ field arithmetic -- + add tree; the extend butterflies and the pointwise Fr stages in 4 x 64-bit Montgomery
arithmetic) on this box's host cores on a bounded sample, and OpenSSL's EC_POINT_mul per point as a third-party datapoint.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PROFILE_TAG = "r06"
LINE_LIMIT = 6000  # the driver keeps an 8 126-character tail of stdout: the ONE line must fit with room to spare
LAUNCHER_VARS = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE",
                 "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS",
                 "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING", "TORCHELASTIC_ERROR_FILE", "OMP_NUM_THREADS")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_share():
    """(threads this process may use, affinity-mask size, cgroup cpu quota or None): os.cpu_count() reports the whole
    host, which a one-GPU box does not own"""
    mask = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = int(q) / int(period)
    except Exception:
        pass
    n = mask if quota is None else max(1, min(mask, int(quota)))
    if os.environ.get("DVP_CPU_THREADS"):
        n = max(1, min(n, int(os.environ["DVP_CPU_THREADS"])))
    return min(n, 256), mask, quota


def cpu_mhz() -> float:
    try:
        vals = [float(l.split(":")[1]) for l in open("/proc/cpuinfo") if l.startswith("cpu MHz")]
        return max(vals) if vals else 0.0
    except Exception:
        return 0.0


def load_profile(name):
    for tag in (PROFILE_TAG, "r05", "r04", "r03"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_{name}.json")))
            d["_profile_tag"] = tag
            return d
        except Exception:
            continue
    return None


def last_json_line(text):
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith("{") and '"metric"' in line:
            try:
                return json.loads(line)
            except Exception:
                pass
    return None


def _sig(x, digits=6):
    """floats to `digits` significant digits, recursively (the line is data: 17-digit floats are noise that costs bytes)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d and d[k] is not None}


def headline(full, detail_path=None):
    """The ONE line the driver parses: the contract's keys first, then short numeric extras.  Everything else this run measured
    (per-round tables, both models' inputs, the CPU baseline's stages, shard plans) is in the sidecar `detail` names."""
    roof = full.get("roofline") or {}
    wm, gm, lr = roof.get("work_model") or {}, roof.get("gather_model") or {}, roof.get("later_rounds") or {}
    cfg = full.get("config") or {}
    cpu = full.get("cpu_baseline")
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    out["config"] = _pick(cfg, ("workload", "log2_constraints", "n_wires", "msm_pairs_per_proof", "witness", "entry", "proofs_in_flight",
                                "ms_per_step_one_shot_msm", "cold_call_s"))
    if isinstance(cfg.get("msm_windows"), dict):
        out["config"]["msm_windows"] = {k: _pick(v, ("c_bits", "windows", "table_gb")) for k, v in cfg["msm_windows"].items()}
    if isinstance(cfg.get("sharding"), str):
        out["config"]["sharding"] = cfg["sharding"][:120]
    out["roofline"] = {"kernel": str(roof.get("kernel", ""))[:64], "bound": roof.get("bound"), "achieved": roof.get("achieved"), "peak": roof.get("peak"),
                       "unit": roof.get("unit"), "frac": roof.get("frac"), "traffic": roof.get("traffic"),
                       "avg_launch_ms": roof.get("avg_launch_ms"), "launches": roof.get("launches"),
                       "algorithmic_bytes_per_launch": roof.get("algorithmic_bytes_per_launch"), "limiter": roof.get("limiter"),
                       "work_model_frac": wm.get("frac"), "gather_model_frac": gm.get("frac"),
                       "later_rounds": _pick(lr, ("ms_per_step", "launches_per_step", "achieved_gb_s", "frac_of_hbm_peak", "work_model_frac"))}
    if roof.get("per_msm"):
        out["roofline"]["per_msm"] = [_pick(p_, ("pairs", "avg_launch_ms", "frac", "work_model_frac")) for p_ in roof["per_msm"]]
    if cpu:
        out["cpu_baseline"] = _pick(cpu, ("value", "unit", "cores", "kind", "end_to_end_s", "composed_value", "msm_points_per_s",
                                          "best_cpu_pippenger_constraints_per_s"))
        out["cpu_baseline"]["sample"] = str(cpu.get("sample", ""))[:400]
    out["stages_ms_per_step"] = full.get("stages_ms_per_step")
    if full.get("msm_standalone"):
        out["msm_standalone"] = full["msm_standalone"]
    by = {}
    for name, d in (full.get("roofline_by_config") or {}).items():
        if name.startswith("config_2"):
            by[name] = _pick(d, ("ms", "achieved_gb_s", "frac_of_hbm_peak"))
        elif name.startswith("config_3"):
            by[name] = {op: _pick(d[op], ("ms", "frac_of_hbm_peak", "work_model_frac")) for op in ("enter", "exit", "extend_x4") if op in d}
            by[name]["round_trip_exact"] = d.get("round_trip_exact")
        elif name.startswith("config_4"):
            by[name] = _pick(d, ("ms_per_step", "achieved_gb_s", "frac_of_hbm_peak"))
            if "extends" in d:
                by[name]["extends"] = _pick(d["extends"], ("ms_per_step", "frac_of_hbm_peak", "work_model_frac"))
    if by:
        out["roofline_by_config"] = by
    out.update(_pick(full, ("hbm_resident_gb", "tables_gb", "ms_per_step_host_witness", "msm_mpoints_per_s", "host_waits_per_proof", "scaling_measured",
                            "rccl_ranks", "backend", "ms_per_step_ranks", "ms_per_step_inproc", "inproc_error", "replicas", "replicas_error", "stages_measured_in")))
    tif = full.get("throughput_two_in_flight")
    if isinstance(tif, dict) and "constraints_per_s" in tif:
        out["throughput_two_in_flight"] = _pick(tif, ("ms_per_proof", "constraints_per_s"))
    if full.get("profiles"):
        out["profiles"] = _pick(full["profiles"], ("tag", "sources_match"))
    out["detail"] = detail_path
    return _sig(out)


def write_detail(full):
    """the sidecar: everything the run measured, as one JSON document.  profiles/<tag>_bench_detail.json is the committed copy (written
    when the tree is writable: the builder's runs), gpurun_out/bench_detail.json travels back from a GPU box"""
    name = None
    single = full.get("n_gpus") == 1 and (full.get("config") or {}).get("log2_constraints") == 20
    suffix = "" if full.get("n_gpus") == 1 else f"_n{full.get('n_gpus')}" + ("_inproc" if "in-library" in str((full.get("config") or {}).get("sharding")) else "")
    targets = [os.path.join(ROOT, "gpurun_out", f"bench_detail{suffix}.json")]
    if single and os.environ.get("DVP_BENCH_WRITE_PROFILE") == "1":
        targets.append(os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_bench_detail.json"))
    for path in targets:
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                json.dump(full, f, indent=1)
            name = os.path.relpath(path, ROOT)
        except OSError as ex:
            log(f"[bench] could not write {path}: {ex}")
    return name


def launch_ranks(args):
    """--gpus N > 1 and no launcher around us: start the ranks as children of a process that has not touched the GPU"""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("[bench] launching ranks:", " ".join(cmd))
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    got = False
    for line in p.stdout:
        if line.lstrip().startswith("{") and '"metric"' in line:
            sys.stdout.write(line)
            sys.stdout.flush()
            got = True
        else:
            sys.stderr.write(line)
    rc = p.wait()
    if rc == 0 and not got:
        log("[bench] the ranks exited cleanly but printed no result line")
        rc = 1
    sys.exit(rc)


def run_inproc_child(args, share_gpu):
    """the in-library multi-GPU figure (dvp_set_devices) for the same device count, from a child process"""
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_VARS}
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--inproc", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--log-m", str(args.log_m), "--no-cpu-baseline"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=float(os.environ.get("DVP_BENCH_INPROC_TIMEOUT", "150")))
    except subprocess.TimeoutExpired:
        return None, "timeout"
    d = last_json_line(r.stdout)
    if r.returncode != 0 or d is None:
        return None, f"rc {r.returncode}: " + (r.stderr.strip().splitlines() or ["no output"])[-1][:300]
    return d["ms_per_step"], None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-m", type=int, default=20, help="log2 of the number of constraints (default 2^20)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed legs after the timed loop (microbenchmarks, stand-alone MSMs, second table flavour)")
    ap.add_argument("--extras", choices=("all", "ubench", "none"), default="all", help="ubench: only the two multiplier / gather microbenchmarks "
                    "(what a counter pass under rocprofv3 --pmc needs: no child process, no extra proofs)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--replicas", action="store_true", help="N > 1: every rank proves its OWN proofs (N independent provers, no data-path collective); "
                    "`value` is then the aggregate constraints/s of the N replicas (weak scaling).  Without the flag the two MSMs of ONE proof are sharded "
                    "over the ranks (strong scaling, the default: BASELINE config #4) and the replicas figure is measured after the timed loop")
    ap.add_argument("--inproc", action="store_true", help="one process, in-library multi-GPU over devices 0..gpus-1 (dvp_set_devices)")
    ap.add_argument("--in-flight", type=int, default=2, help="2 (default): the extras leg after the timed loop also measures throughput with two proofs in flight "
                    "(throughput_two_in_flight); 1 skips it.  `value` is always one proof at a time")
    args = ap.parse_args()

    if args.gpus > 1 and not args.inproc and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)  # never returns; nothing above this line touches the GPU

    import numpy as np
    import torch
    import torch.distributed as dist
    import ctypes as C

    world = 1 if args.inproc else int(os.environ.get("WORLD_SIZE", "1"))
    rank = 0 if args.inproc else int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.inproc else int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs for a one-GPU box (never set by the driver): DVP_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # DVP_DIST_BACKEND=gloo replaces RCCL, so the sharded path can be exercised end to end without a second GPU
    share_gpu = os.environ.get("DVP_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("DVP_DIST_BACKEND", "nccl")
    dev_index = 0 if share_gpu else local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    if not args.inproc and world != args.gpus:
        log(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
        sys.exit(2)
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    dvp = importlib.import_module("dv-pari_amd")
    dvp.check(dvp.lib.dvp_set_device(dev_index), "dvp_set_device")
    log_m = args.log_m
    m = 1 << log_m

    # ---------------- untimed setup: circuit, witness, SRS (same seeds on every rank) ----------------------
    t0 = time.time()
    inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(log_m)
    td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
    pv = dvp.proving.Prover(inst)
    srs = dvp.srs.verifier_runs_setup(pv, inst, td)
    pv.set_srs(srs)
    w_host = dvp.fr.vec([1] + pub + prv)
    assignment = torch.from_numpy(w_host.view(np.int64)).to(dev)
    if rank == 0:
        log(f"[bench] setup m=2^{log_m} n_wires={inst.n_wires} in {time.time() - t0:.1f}s")
    gpu_backend = dvp.distributed.GpuBackend(pv, dev)
    n_dev_inproc = 1
    if args.inproc and args.gpus > 1:
        ids = [0] * args.gpus if share_gpu else list(range(args.gpus))
        dvp.set_devices(ids)
        n_dev_inproc = args.gpus

    stream = torch.cuda.current_stream().cuda_stream

    def step():
        # one GPU: Proof::prove itself (dvp_prove_dev: the drop-in entry with the witness resident; rounds 1-4 timed the PHASED entries
        # here -- begin / msm_partial / challenge / msm_partial / finish, what a rank of a sharded proof calls -- which wait for the
        # host five times per proof where dvp_prove_dev waits once); N ranks: the phased entries with the MSMs sharded
        if args.inproc or world == 1 or args.replicas:
            return pv.prove_dev(assignment.data_ptr(), stream)
        return dvp.distributed.prove_sharded(gpu_backend, assignment, plan_costs=plan_costs)

    plan_costs = None
    if world > 1 and args.replicas:
        dvp.distributed.probe_collectives(dev)
    elif world > 1:  # communicator set-up is not part of a proof: one throw-away exchange even with --warmup 0
        dvp.distributed.probe_collectives(dev)
        # what the shard plan charges an extender rank is MEASURED here (one timed extend, one timed broadcast of an m-vector, one timed
        # MSM slice; max over ranks) instead of typed in: the plan adapts to the real xGMI cost of the vector exchange
        plan_costs = dvp.distributed.measure_plan_costs(gpu_backend, assignment)
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        proof = step()
    # In the timed loop only the dominant kernel (the first pair round of each MSM) is bracketed by HIP events -- the roofline block's
    # live launch time.  The stage breakdown (seven more event scopes per MSM, each a packet of its own that puts ~10 us between two
    # kernels) is measured in a pass of its own after the timed loop (DVP_BENCH_STAGE_EVENTS=1: everything in the timed loop, as in rounds 1-5)
    stage_events_in_loop = os.environ.get("DVP_BENCH_STAGE_EVENTS") == "1"
    dvp.lib.dvp_profile_reset()
    dvp.lib.dvp_profile_enable(1 if stage_events_in_loop else 2)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proof = step()
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    dvp.lib.dvp_profile_enable(0)
    torch.cuda.synchronize()
    # the dominant kernel's figures belong to the TIMED loop: read them before the stage pass resets the slots
    r0_ms, r0_n = C.c_double(0), C.c_uint64(0)
    dvp.check(dvp.lib.dvp_profile_read(b"msm_affine_round0", C.byref(r0_ms), C.byref(r0_n)))
    n_sh = dvp.lib.dvp_profile_round0_shapes(None, None, None, 0)
    r0_shapes = []
    if n_sh > 0:
        a_p, a_ms, a_n = (C.c_uint64 * n_sh)(), (C.c_double * n_sh)(), (C.c_uint64 * n_sh)()
        dvp.lib.dvp_profile_round0_shapes(a_p, a_ms, a_n, n_sh)
        r0_shapes = [(int(a_p[k]), float(a_ms[k]), int(a_n[k])) for k in range(n_sh)]
    stage_steps = args.steps
    if not stage_events_in_loop:
        stage_steps = max(2, min(args.steps, 6))
        dvp.lib.dvp_profile_reset()
        dvp.lib.dvp_profile_enable(1)
        for _ in range(stage_steps):  # (N ranks: a collective pass like the timed one)
            step()
        torch.cuda.synchronize()
        dvp.lib.dvp_profile_enable(0)
    rank_ms = [own_elapsed / args.steps * 1e3]
    dist_info = None
    if world > 1:
        t = torch.tensor([elapsed, own_elapsed], dtype=torch.float64, device=dev)
        allt = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        elapsed = max(float(x[0].item()) for x in allt)
        rank_ms = [float(x[1].item()) / args.steps * 1e3 for x in allt]
        dist_info = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend()}
    if world > 1 and args.replicas:
        dist_info["replicas"] = {"n": world, "constraints_per_s": m * args.steps * world / elapsed, "ms_per_proof_each": rank_ms}
    elif world > 1:
        # every rank's own stage times (HIP events of its own launches) and its slice of the plan, so that a bad plan shows in one run
        mine = {"rank": rank}
        for nm in ("msm_total", "msm_sort", "msm_affine_round0", "msm_affine_rest", "msm_tail", "extend_total"):
            ms_, n_ = C.c_double(0), C.c_uint64(0)
            dvp.check(dvp.lib.dvp_profile_read(nm.encode(), C.byref(ms_), C.byref(n_)))
            mine[nm + "_ms_per_step"] = ms_.value / stage_steps
        allm = [None] * world
        dist.all_gather_object(allm, mine)
        plan_ = dvp.distributed.shard_plan(world, inst.n_wires, m, extend_pairs=plan_costs)
        dist_info["shard_plan"] = [{"rank": r_, "commit_msm_range": list(pa), "k_msm_range": list(pb), "extends": bool(ne),
                                    "pairs": (pa[1] - pa[0]) + (pb[1] - pb[0])} for r_, (pa, pb, ne) in enumerate(plan_)]
        dist_info["shard_plan_costs"] = plan_costs
        dist_info["stages_ms_per_step_by_rank"] = allm

    def prof(name):
        ms, n = C.c_double(0), C.c_uint64(0)
        dvp.check(dvp.lib.dvp_profile_read(name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    try:
        host_waits = {"stream": prof("host_waits_stream")[1] / stage_steps, "side_stream": prof("host_waits_side")[1] / stage_steps}
    except Exception:  # an older build of the library (DVP_LIB A/B runs) has no such counters
        host_waits = None
    acc_ms, acc_n = r0_ms.value, r0_n.value  # the timed loop's own (read before the stage pass)
    rest_ms, _ = prof("msm_affine_rest")
    sort_ms, _ = prof("msm_sort")
    tail_ms, _ = prof("msm_tail")
    msm_ms, msm_n = prof("msm_total")
    ext_ms, ext_n = prof("extend_total")
    plans = [pv.msm_plan(0), pv.msm_plan(1)]
    tables = [pv.msm_table(0), pv.msm_table(1)]
    sizes = [pv.msm_size(0), pv.msm_size(1)]
    free_b, total_b = torch.cuda.mem_get_info(dev)
    hbm_resident_gb = (total_b - free_b) / 1e9

    ms_inproc, inproc_err = None, None
    if world > 1 and not args.replicas:
        # the OTHER way to use N GPUs (what DESIGN 7 recommends for throughput): N independent provers, one per rank, no data-path
        # collective -- measured here, outside the timed loop, between two barriers; max over ranks
        if os.environ.get("DVP_BENCH_NO_REPLICAS") != "1":
            # an extra: a failure on any rank is reported in the line (replicas_error), never raised -- every rank keeps taking part in
            # the barriers and the all-gather below whatever happened to its own proofs
            rep_err = 0.0
            k_rep = max(2, min(args.steps, 10))
            try:
                own_proof = pv.prove_dev(assignment.data_ptr(), stream)  # rebuilds this rank's full fixed-base tables (untimed)
                if own_proof != proof:
                    rep_err = 2.0
            except Exception as ex:
                log(f"[bench] rank {rank}: replicas leg: {ex!r}")
                rep_err = 1.0
            dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            if not rep_err:
                try:
                    for _ in range(k_rep):
                        pv.prove_dev(assignment.data_ptr(), stream)
                    torch.cuda.synchronize()
                except Exception as ex:
                    log(f"[bench] rank {rank}: replicas leg: {ex!r}")
                    rep_err = 1.0
            own_rep = time.perf_counter() - t1
            dist.barrier()
            t = torch.tensor([time.perf_counter() - t1, own_rep, rep_err], dtype=torch.float64, device=dev)
            allt = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            bad_ranks = [r_ for r_, x in enumerate(allt) if float(x[2].item()) != 0.0]
            if bad_ranks:
                dist_info["replicas_error"] = {"ranks": bad_ranks, "codes": [float(allt[r_][2].item()) for r_ in bad_ranks]}
            else:
                rep_elapsed = max(float(x[0].item()) for x in allt)
                dist_info["replicas"] = {"n": world, "proofs_each": k_rep, "constraints_per_s": m * k_rep * world / rep_elapsed,
                                         "ms_per_proof_each": [float(x[1].item()) / k_rep * 1e3 for x in allt]}
    if world > 1:
        # release this rank's GPU before the in-library measurement: every rank drops its prover and leaves the group
        del gpu_backend
        pv.close()
        del assignment
        torch.cuda.empty_cache()
        dist.barrier()
        dist.destroy_process_group()
        dvp.distributed.forget_groups()
        if rank != 0:
            return
        if os.environ.get("DVP_BENCH_NO_INPROC") != "1" and not args.replicas:
            time.sleep(1.0)  # the other ranks are exiting: their HBM comes back with their processes
            ms_inproc, inproc_err = run_inproc_child(args, share_gpu)
            if inproc_err:
                log(f"[bench] in-library multi-GPU leg failed: {inproc_err}")

    # correctness of what was timed: the proof must verify
    assert dvp.srs.verify(td, pub, proof), "bench proof does not verify"
    n_shards = (1 if args.replicas else world) * n_dev_inproc
    ms_per_step = elapsed / args.steps * 1e3
    value = m * args.steps / elapsed * (world if (args.replicas and world > 1) else 1)
    single = world == 1 and n_dev_inproc == 1
    # `value` / `ms_per_step` are ONE PROOF AT A TIME: the quantity BASELINE's metric measures (the reference's prove() is one proof
    # per call), and the loop that stages_ms_per_step, the roofline block and every committed profile describe.  Throughput with
    # two proofs in flight is an extra, measured after the timed loop and reported under its own key (throughput_two_in_flight).
    n_in_flight = 1
    extras = single and not args.no_extras and args.extras != "none"
    heavy = extras and args.extras == "all"  # the legs that run further proofs / MSMs / a child process

    # ---- outside the timed region ---------------------------------------------------------------------------------------
    host_ms = mul_rate = gather_rate = fr_rate = ms_one_shot = cold = ecfft_live = None
    msm_standalone = None
    in_flight = None
    if extras:
        # the two ceilings of the dominant kernel: the multiplier alone, and random 64-byte gathers out of the table the
        # K-MSM's first pair round reads (the larger of the two tables)
        r = C.c_double(0)
        dvp.check(dvp.lib.dvp_ubench_gf_mul(200, C.byref(r)), "dvp_ubench_gf_mul")
        mul_rate = r.value
        tp, tb = C.c_void_p(0), C.c_uint64(0)
        dvp.check(dvp.lib.dvp_prover_msm_table_ptr(pv._h, 1, C.byref(tp), C.byref(tb)), "dvp_prover_msm_table_ptr")
        if tp.value and tb.value >= (1 << 20):
            dvp.check(dvp.lib.dvp_ubench_gather(tp, tb.value, 4, C.byref(r)), "dvp_ubench_gather")
            gather_rate = r.value
        # the ECFFT's multiplier alone (ceiling of the work model of extend / enter / exit)
        dvp.check(dvp.lib.dvp_ubench_fr_mul(400, C.byref(r)), "dvp_ubench_fr_mul")
        fr_rate = r.value
    if heavy:
        # the same proof through the host-pointer seam (dvp_prove = Proof::prove's signature, src/proving.rs:426: witness in
        # host memory, +32 B/wire of H2D)
        pv.prove(pub, prv)
        reps = max(3, args.steps // 2)
        pub_l, prv_l = w_host[1:1 + len(pub)], w_host[1 + len(pub):]
        host_times = []
        for _ in range(reps):
            t1 = time.perf_counter()
            p2 = pv.prove(pub_l, prv_l)
            host_times.append((time.perf_counter() - t1) * 1e3)
        # median: the 32 MB copy out of pageable host memory now and then takes tens of ms (one call in ~50: the runtime's staging),
        # which says nothing about the seam
        host_ms = sorted(host_times)[len(host_times) // 2]
        assert p2 == proof
        # BASELINE metric, second half: stand-alone one-shot MSM (multi_scalar_mul, src/curve.rs:141-158; no pre-rotated
        # tables), device-resident random scalars x real SRS bases; config #2 is the 2^16 case
        if log_m >= 18:
            gk = srs.as_list()[2:]
            bases_np = np.ascontiguousarray(np.concatenate([np.asarray(x[0], dtype=np.uint64) for x in gk]))  # 4m bases
            d_bases = torch.from_numpy(bases_np.view(np.int64)).to(dev)
            rng = np.random.default_rng(2)
            sc = rng.integers(0, 2**62, size=(bases_np.shape[0], 4), dtype=np.uint64)
            sc[:, 3] &= np.uint64((1 << 38) - 1)
            d_sc = torch.from_numpy(sc.view(np.int64)).to(dev)
            d_out = torch.zeros(10, dtype=torch.int64, device=dev)
            msm_standalone = {}
            for lg in (16, 20, 22):
                n_pts = 1 << lg
                if n_pts > bases_np.shape[0]:
                    continue
                best = None
                for it in range(4):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    dvp.check(dvp.lib.dvp_msm_affine_dev(d_sc.data_ptr(), d_bases.data_ptr(), None, n_pts, d_out.data_ptr(), d_out.data_ptr() + 64, stream))
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t1
                    if it and (best is None or dt < best):
                        best = dt
                msm_standalone[f"2^{lg}"] = {"ms": best * 1e3, "mpoints_per_s": n_pts / best / 1e6}
            del d_bases, d_sc
        # BASELINE config #3, live: 2^20-coefficient enter / exit round trip and the prover's own op, extend (x4 vectors, m = 2^20)
        if log_m >= 20:
            n3 = 1 << 20
            t3 = dvp.ec_fft.FFTree(n3)
            rng3 = np.random.default_rng(3)
            c3 = rng3.integers(0, 2**62, size=(n3, 4), dtype=np.uint64)
            c3[:, 3] &= np.uint64((1 << 38) - 1)
            d_in = torch.from_numpy(c3.view(np.int64)).to(dev)
            d_ev, d_back = torch.empty_like(d_in), torch.empty_like(d_in)

            def timeit(f, reps=3):
                f()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(reps):
                    f()
                torch.cuda.synchronize()
                return (time.perf_counter() - t1) / reps * 1e3
            ms_enter = timeit(lambda: t3.enter_dev(d_in.data_ptr(), d_ev.data_ptr(), stream))
            t3.exit_dev(d_ev.data_ptr(), d_back.data_ptr(), stream)  # first call bootstraps the exit tables
            ms_exit = timeit(lambda: t3.exit_dev(d_ev.data_ptr(), d_back.data_ptr(), stream))
            round_trip_exact = bool((d_back == d_in).all().item())
            t3.close()
            t6 = dvp.ec_fft.FFTree(2 * n3)
            x4 = d_in.reshape(1, n3, 4).repeat(4, 1, 1).contiguous()
            y4 = torch.empty_like(x4)
            ms_ext4 = timeit(lambda: t6.extend_dev(x4.data_ptr(), 4, y4.data_ptr(), stream))
            t6.close()
            del d_in, d_ev, d_back, x4, y4
            ecfft_live = {"n": n3, "enter_ms": ms_enter, "exit_ms": ms_exit, "extend_x4_ms": ms_ext4, "round_trip_exact": round_trip_exact}
            assert round_trip_exact, "exit(enter(c)) != c"
        # the same proof WITHOUT the fixed-base tables (one-shot MSMs over the decoded bases: what a prover that keeps nothing but the
        # SRS between calls would run)
        with dvp.tune(DVP_MSM_FIXED_MIN=1 << 40):
            assert pv.prove_dev(assignment.data_ptr(), stream) == proof, "one-shot-MSM proof differs"
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                pv.prove_dev(assignment.data_ptr(), stream)
            torch.cuda.synchronize()
            ms_one_shot = (time.perf_counter() - t1) / 3 * 1e3
        # the COLD call: Proof::prove(cache_dir, ..) from a fresh process (tools/cold_call.py) on a directory written here with the
        # same trapdoor -- R1CS dump + five SRS point files read, 6 m points decoded, tables built, one proof
        cold = None
        if os.environ.get("DVP_BENCH_NO_COLD") != "1":
            import shutil
            import tempfile
            tmp = tempfile.mkdtemp(prefix="dvp_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
            try:
                inst.write_dump_file(os.path.join(tmp, dvp.artifacts.R1CS_CONSTRAINTS_FILE))
                nat = importlib.import_module("dv-pari_amd._native")
                t_, d_, e_ = (dvp.fr.limbs(x) for x in (td.tau, td.delta, td.epsilon))
                t1 = time.perf_counter()
                dvp.check(dvp.lib.dvp_setup_cache_dir(nat.ptr(t_), nat.ptr(d_), nat.ptr(e_), os.fsencode(tmp), len(pub), 0), "dvp_setup_cache_dir")
                setup_s = time.perf_counter() - t1
                np.save(os.path.join(tmp, "witness.npy"), w_host)
                # not a second profiled process: a profiler's preload would make the child collect counters while this one holds the GPU
                env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_VARS and k != "LD_PRELOAD" and not k.startswith(("ROCP_", "ROCPROF"))}
                rr = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cold_call.py"), tmp, os.path.join(tmp, "witness.npy"), str(len(pub))],
                                    capture_output=True, text=True, env=env, timeout=240)
                lines = [l for l in rr.stdout.splitlines() if l.startswith("{")]
                if rr.returncode == 0 and lines:
                    cold = json.loads(lines[-1])
                    cold["same_bytes_as_timed_proof"] = cold.pop("proof_hex") == proof.to_bytes().hex()
                    cold["setup_cache_dir_s"] = setup_s
                    cold["files_mb"] = round(sum(os.path.getsize(os.path.join(tmp, f)) for f in os.listdir(tmp) if f != "witness.npy") / 1e6, 1)
                    assert cold["same_bytes_as_timed_proof"], "the cold dvp_prove_cache_dir proof differs from the timed loop's"
                else:
                    cold = {"error": (rr.stderr.strip().splitlines() or ["no output"])[-1][:300]}
            except Exception as ex:  # the leg is an extra: report, do not fail the line
                cold = {"error": repr(ex)[:300]}
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
        # throughput with TWO proofs in flight on this GPU: a second prover (own tables, own stream, own host thread) over the same
        # circuit; the library lets the two MSMs overlap everything but their pair rounds (msm.hip: HeavyGate).  Every proof is
        # compared with the timed loop's bytes.  Not the headline: `value` stays one proof at a time.  The leg is skipped only on a
        # PRE-CHECKED capacity condition (no room for a second prover); any library / HIP error inside it fails the run.
        free_b2, total_b2 = torch.cuda.mem_get_info(dev)
        if args.in_flight < 2:
            in_flight = None
        elif free_b2 <= 1.15 * (total_b2 - free_b2):
            in_flight = {"skipped": f"no room for a second prover: {free_b2 / 1e9:.1f} GB free, {(total_b2 - free_b2) / 1e9:.1f} GB in use by the first"}
        else:
            import threading
            pv_b = dvp.proving.Prover(inst)
            try:
                pv_b.set_srs(srs)
                st_b = torch.cuda.Stream()
                pair = ((pv, stream), (pv_b, st_b.cuda_stream))
                k_each = max(4, min(args.steps, 12))
                bad, errs = [], []

                def _loop(pvx, stx, k):
                    try:
                        for _ in range(k):
                            if pvx.prove_dev(assignment.data_ptr(), stx) != proof:
                                bad.append(1)
                    except Exception as e:  # an exception in a thread would otherwise vanish and leave a short, wrong timing
                        errs.append(repr(e))
                for k in (2, k_each):  # first pass = concurrent warm-up (the second MSM workspace is allocated on first overlap)
                    th = [threading.Thread(target=_loop, args=(a, b, k)) for a, b in pair]
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for t in th:
                        t.start()
                    for t in th:
                        t.join()
                    torch.cuda.synchronize()
                    dt2 = time.perf_counter() - t1
                    if errs:
                        raise RuntimeError("two-in-flight loop: " + "; ".join(errs))
                assert not bad, "a proof computed with two in flight differs"
                free_b3, total_b3 = torch.cuda.mem_get_info(dev)
                in_flight = {"provers": 2, "proofs": 2 * k_each, "ms_per_proof": dt2 / (2 * k_each) * 1e3,
                             "constraints_per_s": m * 2 * k_each / dt2, "hbm_resident_gb": (total_b3 - free_b3) / 1e9,
                             }
            finally:
                pv_b.close()


    pairs_total = (inst.n_wires + m + 4 * m) * args.steps / n_shards  # (scalar, base) pairs this rank pushed through the kernel
    pairs_per_launch = pairs_total / max(acc_n, 1)
    acc_avg_ms = acc_ms / max(acc_n, 1)
    alg_bytes = 96.0 * pairs_per_launch
    achieved = alg_bytes / (acc_avg_ms * 1e-3) / 1e9 if acc_n else 0.0

    # entries per scalar: one per window (the signed aligned windows: ceil(234 / c); the tau-adic ones the same over ~234 digits)
    def per_scalar(c, _signed):
        return float(-(-234 // max(c, 1)))
    w_eff = (sum(per_scalar(c, sg) * n for (c, _), (_, sg), n in zip(plans, tables, sizes)) / max(sum(sizes), 1)
             if all(c for c, _ in plans) else 12)
    # per addition: 5 products + 1 squaring (~0.13 product) + 1/B of a table-driven inversion (~15 product-equivalents);
    # B = slots per thread, chosen on the device so that a round is whole chip-fulls (msm.hip: aff_slots_per_thread): restated here
    def tune_get(name, dflt):
        v = C.c_longlong(0)
        try:
            return int(v.value) if dvp.lib.dvp_tune_get(name.encode(), C.byref(v)) == 0 and v.value > 0 else dflt
        except Exception:
            return dflt
    lib_bmax, lib_bmin = tune_get("DVP_MSM_AFF_BMAX", 136), tune_get("DVP_MSM_AFF_BMIN", 8)

    def aff_b(total_slots, cap=256 * 3 * 256, bmax=lib_bmax, bmin=lib_bmin):
        units = max(1, -(-int(total_slots) // cap))
        r = -(-units // bmax)
        return max(bmin, -(-units // r))
    adds_per_launch = pairs_per_launch * w_eff * 0.5
    shape_adds = [n * per_scalar(c, True) * 0.5 / n_shards for (c, _), n in zip(plans, sizes)] if all(c for c, _ in plans) else [adds_per_launch]
    per_add = sum(a * (5.13 + 15.0 / aff_b(a)) for a in shape_adds) / max(sum(shape_adds), 1.0)
    mul_eq = adds_per_launch * per_add
    launch_s = acc_avg_ms * 1e-3
    mul_frac = (mul_eq / launch_s) / mul_rate if (acc_n and mul_rate) else None
    gathers = 4.0 * adds_per_launch  # 64-byte lines: both operands of an addition, once per pass
    gather_frac = (gathers / launch_s) / gather_rate if (acc_n and gather_rate) else None
    # "bound" names the peak the contract's achieved / peak / frac triple is quoted against (algorithmic bytes against HBM streaming
    # bandwidth); "limiter" is what the two models measured in this run say actually binds the kernel
    bound = "hbm"
    if gather_frac is not None and mul_frac is not None:
        limiter = "gather_rate" if gather_frac >= mul_frac else "valu_lds"
    else:
        limiter = None
    roof = {
        "kernel": "dvp::k_affine_round<true>",
        "bound": bound,
        "limiter": limiter,
        "achieved": achieved,
        "peak": 8000.0,
        "unit": "GB/s",
        "frac": achieved / 8000.0,
        "traffic": None,
        "launches": int(acc_n),
        "avg_launch_ms": acc_avg_ms,
        "algorithmic_bytes_per_launch": alg_bytes,
        "gather_model": {
            "gathers_per_launch": gathers,
            "achieved_gathers_per_s": gathers / launch_s if acc_n else 0.0,
            "ceiling_gathers_per_s": gather_rate,
            "table_gb": round(tables[1][0] / 1e9, 2),
            "frac": gather_frac,
        },
        "work_model": {
            "product_equivalents_per_addition": per_add,
            "additions_per_launch": adds_per_launch,
            "mul_equivalents_per_launch": mul_eq,
            "achieved_mul_per_s": mul_eq / launch_s if acc_n else 0.0,
            "multiplier_microbench_mul_per_s": mul_rate,
            "frac": mul_frac,
        },
    }
    # the same triple per launch SHAPE (the commit MSM and the K MSM are different sizes; the block above is their average)
    if r0_shapes:  # (the timed loop's launches, read before the stage pass)
        per = []
        for sh_pairs, sh_ms, sh_n in r0_shapes:
            pairs_k, ms_k = float(sh_pairs), sh_ms / max(sh_n, 1)
            adds_k = pairs_k * w_eff * 0.5
            per.append({"pairs": int(sh_pairs), "launches": int(sh_n), "avg_launch_ms": ms_k,
                        "achieved_gb_s": 96.0 * pairs_k / (ms_k * 1e-3) / 1e9, "frac": 96.0 * pairs_k / (ms_k * 1e-3) / 1e9 / 8000.0,
                        "additions_per_launch": adds_k,
                        "work_model_frac": (adds_k * per_add / (ms_k * 1e-3)) / mul_rate if mul_rate else None})
        roof["per_msm"] = per
    traffic = load_profile("pmc_traffic_k_affine_round0")
    if traffic and log_m == 20 and n_shards == 1:
        raw = traffic["traffic_bytes_per_launch_raw"]
        roof["traffic"] = traffic.get("traffic_bytes_per_launch_best", traffic["traffic_bytes_per_launch_fetch_x2"])
        roof["traffic_detail"] = {
            "source": f"profiles/{traffic['_profile_tag']}_pmc_traffic_k_affine_round0.json (committed rocprofv3 --pmc passes of this command; not measured in this run)",
            "raw_bytes_per_launch": raw,
            "fetch_x2_bytes_per_launch": traffic["traffic_bytes_per_launch_fetch_x2"],
            "ratio_to_algorithmic": roof["traffic"] / alg_bytes,
        }
        for k in ("tcc_requests",):
            if k in traffic:
                roof["traffic_detail"][k] = traffic[k]
    sq = load_profile("pmc_sq_k_affine_round0")
    if sq and log_m == 20 and n_shards == 1:
        roof["issue"] = sq
    # the OTHER half of the pair-round time: k_affine_round<false>, the later rounds (same code, coalesced inputs).  Live: HIP-event
    # total and launches of this run; per round (additions, bytes, clock, issue rate): the committed request-level / SQ passes
    rounds_prof = load_profile("pmc_pair_rounds_by_round")
    later_launches = prof("msm_affine_rest")[1]

    def later_additions(pairs, c):  # rounds 1 .. : round r adds entries / 2^(r+1) pairs, and runs while that is >= 2^19 (Tune::msm_aff_min)
        e, tot, work, r = pairs * per_scalar(c, True) / n_shards, 0.0, 0.0, 1
        while e / 2 ** (r + 1) >= (1 << 19):
            a = e / 2 ** (r + 1)
            tot += a
            work += a * (5.13 + 15.0 / aff_b(a))  # product equivalents: the inversion is shared by the B slots of a thread
            r += 1
        return tot, work
    later = [later_additions(n, c) for (c, _), n in zip(plans, sizes)] if all(c for c, _ in plans) else []
    later_adds = sum(a for a, _ in later)  # per proof
    later_work = sum(w for _, w in later)
    later_s = rest_ms / stage_steps * 1e-3
    roof["later_rounds"] = {
        "kernel": "dvp::k_affine_round<false>",
        "ms_per_step": rest_ms / stage_steps, "launches_per_step": later_launches / stage_steps,
        "additions_per_step": later_adds,
        "algorithmic_bytes_per_addition": 128.0,
        "achieved_gb_s": later_adds * 128.0 / later_s / 1e9 if later_s else None,
        "frac_of_hbm_peak": later_adds * 128.0 / later_s / 1e9 / 8000.0 if later_s else None,
        "work_model_frac": (later_work / later_s) / mul_rate if (mul_rate and later_s) else None,
        "product_equivalents_per_addition": later_work / later_adds if later_adds else None,
    }
    if rounds_prof and log_m == 20 and n_shards == 1:
        roof["later_rounds"]["by_round"] = [
            {"msm": msm["which"], **{k: r[k] for k in ("round", "kernel", "additions", "ms_under_pmc", "additions_per_s", "read_requests_per_addition",
                                                         "traffic_bytes_per_addition", "traffic_tb_per_s", "l2_hit_rate", "effective_clock_ghz",
                                                         "valu_insts_per_simd_cycle", "resident_wave_frac", "valu_insts_per_addition")}}
            for msm in rounds_prof["msms"] for r in msm["rounds"]]
        roof["later_rounds"]["by_round_source"] = f"profiles/{rounds_prof['_profile_tag']}_pmc_pair_rounds_by_round.json (committed rocprofv3 --pmc passes; not measured in this run)"
    g64 = load_profile("gather64_load_forms")
    if g64:
        roof["request_size_experiment"] = {"source": f"profiles/{g64['_profile_tag']}_gather64_load_forms.json (tools/ubench/gather64.hip under rocprofv3 --pmc)",
                                           "fabric_read_bytes_per_64_byte_point": {v["variant"]: round(v["fabric_read_bytes_per_point"], 1) for v in g64["variants"]}}
    out = {
        "metric": "R1CS constraints/sec (prove)",
        "value": value,
        "unit": "constraints/s",
        "n_gpus": args.gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak" if (args.replicas and world > 1) else "strong",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {
            "workload": f"Proof::prove, synthetic dense R1CS, 2^{log_m} constraints" + (" (BASELINE config #4)" if log_m == 20 else ""),
            "log2_constraints": log_m,
            "n_wires": inst.n_wires,
            "msm_pairs_per_proof": inst.n_wires + 5 * m,
            "sharding": ("in-library (dvp_set_devices): MSM index ranges per device, one host thread each, partial points added on device 0"
                         if n_dev_inproc > 1 else
                         "independent replicas: one whole proof per rank per step, no data-path collective" if (args.replicas and world > 1) else
                         "MSM index ranges per rank, all-gather of partial points + local add; extends by vector among the ranks that need q2 / r2 "
                         "(from three such ranks up: one broadcast per vector); challenge "
                         "phase (inversions, barycentric sums, K scalars) by index with one all-gather of 128-byte records" if world > 1 else "single GPU"),
            "msm_windows": {"commit_msm": {"c_bits": plans[0][0], "windows": plans[0][1], "signed_windows": tables[0][1], "table_gb": round(tables[0][0] / 1e9, 2)},
                            "k_msm": {"c_bits": plans[1][0], "windows": plans[1][1], "signed_windows": tables[1][1], "table_gb": round(tables[1][0] / 1e9, 2)}},
            "witness": "resident in HBM",
            "entry": "dvp_prove_dev (Proof::prove with the witness resident)" if (args.inproc or world == 1 or args.replicas) else "phased entries, MSMs sharded over the ranks",
            "proofs_in_flight": 1,
            "latency_ms_one_proof": ms_per_step,
            "ms_per_step_one_shot_msm": ms_one_shot,
            "cold_call_s": (cold or {}).get("cold_call_s"),
            "cold_call": cold,
            "constraints_per_s_two_in_flight": (in_flight or {}).get("constraints_per_s"),
        },
        "hbm_resident_gb": hbm_resident_gb,
        "tables_gb": round((tables[0][0] + tables[1][0]) / 1e9, 2),
        "ms_per_step_host_witness": host_ms,
        "throughput_two_in_flight": in_flight,
        "host_waits_per_proof": host_waits,
        "roofline": roof,
        "stages_ms_per_step": {
            "msm_total": msm_ms / stage_steps,
            "msm_recode_sort": sort_ms / stage_steps,
            "msm_affine_round0": acc_ms / args.steps,  # the timed loop's own events
            "msm_affine_later_rounds": rest_ms / stage_steps,
            "msm_merge_frobenius_tail": tail_ms / stage_steps,
            "extend": ext_ms / stage_steps,
        },
        "stages_measured_in": ("the timed loop" if stage_events_in_loop else
                               f"a pass of {stage_steps} proofs after the timed loop (msm_affine_round0: the timed loop's own events)"),
        "msm_mpoints_per_s": (pairs_total / args.steps * stage_steps / (msm_ms * 1e-3) / 1e6) if msm_ms else None,
        "msm_standalone": msm_standalone,
    }
    # ---- every BASELINE config against its own algorithmic bytes, so that a reader of this line and profiles/ can recompute a fraction
    by_cfg = {}
    if msm_standalone and "2^16" in msm_standalone:
        t16 = msm_standalone["2^16"]["ms"] * 1e-3
        by_cfg["config_2_msm_2p16"] = {"ms": msm_standalone["2^16"]["ms"], "algorithmic_bytes": 96.0 * 65536, "achieved_gb_s": 96.0 * 65536 / t16 / 1e9,
                                       "frac_of_hbm_peak": 96.0 * 65536 / t16 / 1e9 / 8000.0, "bound": "latency (a chain of ~40 dependent launches; 6.3 MB of input)",
                                       "profile": f"profiles/{PROFILE_TAG}_msm_2p16_kernel_stats.csv"}
    if ecfft_live:
        n3 = ecfft_live["n"]
        lg3 = n3.bit_length() - 1
        ent_b, ext_b, x4_b = 320.0 * n3 * lg3, 640.0 * n3 * lg3, (64.0 * 4 + 256.0) * n3
        ent_w, x4_w = n3 * lg3 * lg3 / 2.0 * 2, 4 * 2.0 * n3 * lg3  # multiply-adds: an extend of m values is 2 m log2(m); enter(n) = sum over levels of 2 extends of half size
        by_cfg["config_3_ecfft_2p20"] = {
            "enter": {"ms": ecfft_live["enter_ms"], "algorithmic_bytes": ent_b, "achieved_gb_s": ent_b / (ecfft_live["enter_ms"] * 1e-3) / 1e9,
                      "frac_of_hbm_peak": ent_b / (ecfft_live["enter_ms"] * 1e-3) / 1e9 / 8000.0,
                      "work_model_frac": (ent_w / (ecfft_live["enter_ms"] * 1e-3)) / fr_rate if fr_rate else None},
            "exit": {"ms": ecfft_live["exit_ms"], "algorithmic_bytes": ext_b, "achieved_gb_s": ext_b / (ecfft_live["exit_ms"] * 1e-3) / 1e9,
                     "frac_of_hbm_peak": ext_b / (ecfft_live["exit_ms"] * 1e-3) / 1e9 / 8000.0,
                     "work_model_frac": (2 * ent_w / (ecfft_live["exit_ms"] * 1e-3)) / fr_rate if fr_rate else None},
            "extend_x4": {"ms": ecfft_live["extend_x4_ms"], "algorithmic_bytes": x4_b, "achieved_gb_s": x4_b / (ecfft_live["extend_x4_ms"] * 1e-3) / 1e9,
                          "frac_of_hbm_peak": x4_b / (ecfft_live["extend_x4_ms"] * 1e-3) / 1e9 / 8000.0,
                          "work_model_frac": (x4_w / (ecfft_live["extend_x4_ms"] * 1e-3)) / fr_rate if fr_rate else None},
            "round_trip_exact": ecfft_live["round_trip_exact"],
            "fr_multiplier_muladds_per_s": fr_rate, "profile": f"profiles/{PROFILE_TAG}_config3_ecfft_2p20_kernel_stats.csv"}
    if ext_ms and fr_rate:
        n_ext_v = 3
        by_cfg["config_4_prove_2p%d" % log_m] = {
            "ms_per_step": ms_per_step, "algorithmic_bytes_per_constraint": 2400.0, "achieved_gb_s": 2400.0 * m / (ms_per_step * 1e-3) / 1e9,
            "frac_of_hbm_peak": 2400.0 * m / (ms_per_step * 1e-3) / 1e9 / 8000.0,
            "extends": {"vectors": n_ext_v, "ms_per_step": ext_ms / stage_steps, "algorithmic_bytes": (64.0 * n_ext_v + 256.0) * m,
                        "frac_of_hbm_peak": (64.0 * n_ext_v + 256.0) * m / (ext_ms / stage_steps * 1e-3) / 1e9 / 8000.0,
                        "work_model_frac": (n_ext_v * 2.0 * m * log_m / (ext_ms / stage_steps * 1e-3)) / fr_rate},
            "bound": "GF(2^233) products of the two MSMs' pair rounds (roofline.work_model / roofline.later_rounds)"}
    for key, name in (("config_5_sparse_2p22", "config5_sparse_2p22"), ("setup_2p20", "setup_2p20")):
        try:
            txt = [l.strip() for l in open(os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_{name}.log")) if l.strip()]
            by_cfg[key] = {"source": f"profiles/{PROFILE_TAG}_{name}.log + _kernel_stats.csv (committed rocprofv3 --kernel-trace --stats run; not measured in this run)",
                           "result_lines": [l for l in txt if ("ms per proof" in l or "dvp_setup_cache_dir" in l or "prepares_precomputes" in l or l.startswith("setup ") or "rows" in l)][:8]}
        except Exception:
            pass
    out["roofline_by_config"] = by_cfg
    # ---- the committed profiles: taken at which sources?
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import profile_stamp
        cur = profile_stamp.source_sha16(ROOT)
        st_ = json.load(open(os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_profile_stamp.json")))
        out["profiles"] = {"tag": PROFILE_TAG, "profile_source_sha16": st_["source_sha16"], "current_source_sha16": cur,
                           "sources_match": st_["source_sha16"] == cur, "profile_commit": st_.get("git_head_when_digested"),
                           "warning": None if st_["source_sha16"] == cur else "the committed counters (roofline.traffic / issue / later_rounds.by_round) were taken at OTHER kernel sources than this run's"}
    except Exception as ex:
        out["profiles"] = {"tag": PROFILE_TAG, "warning": f"no profile stamp: {ex!r}"[:200]}
    out["scaling_measured"] = bool(world > 1 or n_dev_inproc > 1)
    if dist_info:
        out.update(dist_info)
        out["ms_per_step_ranks"] = rank_ms
        out["ms_per_step_inproc"] = ms_inproc
        if inproc_err:
            out["inproc_error"] = inproc_err

    if single and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import c_oracle as co

        cores, mask, quota = cpu_share()
        mhz = cpu_mhz()
        rng = np.random.default_rng(99)

        def rand(n):
            s = rng.integers(0, 2**62, size=(n, 4), dtype=np.uint64)
            s[:, 3] &= np.uint64((1 << 38) - 1)
            return s

        pts_per_constraint = (inst.n_wires + 5 * m) / m
        probe = 1 << 13
        bases = np.ascontiguousarray(srs.g_k[2][0])  # 2m real SRS bases
        cap = bases.shape[0]
        sc = rand(cap)
        t1 = time.perf_counter()
        co.msm(sc[:probe], bases[:probe], threads=cores)
        rate = probe / (time.perf_counter() - t1)
        n_s = int(min(cap, max(probe, rate * args.cpu_seconds)))
        t1 = time.perf_counter()
        co.msm(sc[:n_s], bases[:n_s], threads=cores)
        dt = time.perf_counter() - t1
        pts_per_s = n_s / dt
        us_core = cores / pts_per_s * 1e6
        # the four extends of a proof (src/proving.rs:410-422): 2 log2(m) butterfly passes each, 4 Fr products per pair, in the
        # 4 x 64-bit Montgomery arithmetic of the reference's Fr; timing is data-independent, so one vector with synthetic
        # matrices is timed and scaled
        ext_n = 1 << min(log_m, 18)
        data = rand(ext_n)
        mats = np.ascontiguousarray(rand(2 * (ext_n // 2) * 4).reshape(2, ext_n // 2, 4, 4))
        passes = 2 * (ext_n.bit_length() - 1)
        t1 = time.perf_counter()
        co.fr_butterfly_passes(data, mats, passes, cores)
        dt_ext = time.perf_counter() - t1
        ns_per_frmul = dt_ext / (passes * ext_n * 2) * 1e9 * cores
        ext_s_per_proof = 4 * (2 * log_m) * (m * 2) * ns_per_frmul * 1e-9 / cores
        # the pointwise stages (src/proving.rs:492-654: quotient, three barycentric evaluations with their own batch inversions,
        # denominators, K scalars) actually run on 2^18-element vectors and scaled linearly; the reference's sparse mat-vec
        # and its barycentric loops are sequential, which this figure does not charge
        pw_n = 1 << min(log_m, 18)
        pointwise_s_per_proof = co.fr_pointwise_stages(pw_n, cores) * (m / pw_n)
        msm_s_per_proof = (inst.n_wires + 5 * m) / pts_per_s
        cpu_s = msm_s_per_proof + ext_s_per_proof + pointwise_s_per_proof
        # "best CPU" (BASELINE.md B3): a host-side bucket method on the same cores
        n_p = int(min(cap, max(1 << 16, 4 * n_s)))
        t1 = time.perf_counter()
        co.msm_pippenger(sc[:n_p], bases[:n_p], threads=cores)
        pip_pts_per_s = n_p / (time.perf_counter() - t1)
        ossl = None
        try:
            n_o = max(64 * cores, 256)
            t1 = time.perf_counter()
            r = co.openssl_msm(sc[:n_o], bases[:n_o], threads=cores)
            if r is not NotImplemented:
                ossl = n_o / (time.perf_counter() - t1)
        except Exception as e:  # the third-party datapoint is optional
            log(f"[bench] OpenSSL datapoint unavailable: {e}")
        # ---- a real CPU Proof::prove, end to end (oracle/dvp_oracle.c: dvo_prove_commit / dvo_prove_open, the reference's shape
        # incl. its SEQUENTIAL mat-vec and barycentric loops) at 2^16 (and 2^18) constraints: bytes compared with the GPU prover's
        # proof of the same instance, measured seconds beside the figure composed from the samples above at that size
        import pyref as oref

        def cpu_end_to_end(log_e):
            me = 1 << log_e
            own = log_e != log_m  # at the bench size the timed prover, its circuit and its SRS are reused
            if own:
                inst_e, pub_e, prv_e = dvp.gnark_r1cs.synthetic_dense(log_e)
                pv_e = dvp.proving.Prover(inst_e)
            else:
                inst_e, pub_e, prv_e, pv_e = inst, pub, prv, pv
            try:
                if own:
                    srs_e = dvp.srs.verifier_runs_setup(pv_e, inst_e, td)
                    pv_e.set_srs(srs_e)
                else:
                    srs_e = srs
                gpu_proof = pv_e.prove(pub_e, prv_e)
                d_e, d2_e = pv_e.domains()
                bar_e, z2inv_e = pv_e.domain_tables(0)
                tree_e = dvp.ec_fft.FFTree(2 * me)
                z_poly_e = dvp.ec_fft.compute_vanishing_polynomial(tree_e, 0)
                mats = []
                for which in (0, 1):
                    a_ = np.zeros(((me - 1) * 4, 4), dtype=np.uint64)
                    dvp.check(dvp.lib.dvp_debug_ecfft_matrices(tree_e._h, 0, which, a_.ctypes.data_as(C.c_void_p)))
                    mats.append(a_)
                tree_e.close()
                lst = srs_e.as_list()
                bases_a = np.concatenate([np.asarray(lst[0][0], dtype=np.uint64), np.asarray(lst[1][0], dtype=np.uint64)])
                bases_k = np.concatenate([np.asarray(x[0], dtype=np.uint64) for x in lst[2:]])
                csr = [(mt.row_ptr, mt.wire, mt.coeff) for mt in (inst_e.l, inst_e.r, inst_e.o)]
                inp = co.ProveInputs(me, inst_e.n_wires, len(pub_e), csr, inst_e.coeffs, inst_e.n_rows, d_e, d2_e, bar_e, z2inv_e, z_poly_e,
                                     mats[0], mats[1], bases_a, bases_k)
                t_e = time.perf_counter()
                commit, kzg, a0, b0, stages = co.prove_cpu(inp, pub_e, prv_e, lambda cm: oref.transcript_challenge(cm, pub_e), threads=cores)
                wall = time.perf_counter() - t_e
                same = (commit == gpu_proof.commit_p and kzg == gpu_proof.kzg_k and a0.to_bytes(29, "little") == gpu_proof.a0
                        and b0.to_bytes(29, "little") == gpu_proof.b0)
                assert same, f"CPU end-to-end proof at 2^{log_e} differs from the GPU prover's bytes"
                e2e = sum(stages.values())
                composed = ((inst_e.n_wires + 5 * me) / pts_per_s + 4 * (2 * log_e) * (me * 2) * ns_per_frmul * 1e-9 / cores
                            + pointwise_s_per_proof * (me / m))
                return {"log2_constraints": log_e, "end_to_end_s": e2e, "wall_s_incl_input_conversion": wall, "stages_s": stages,
                        "constraints_per_s": me / e2e, "composed_s_at_this_size": composed, "measured_over_composed": e2e / composed,
                        "bytes_equal_gpu_proof": True}
            finally:
                if own:
                    pv_e.close()

        # 2^16 (the composition's cross-check at a small size) and THE BENCH SIZE ITSELF: cpu_baseline.value is the measured figure there
        e2e_sizes = sorted(set(([16] if log_m > 16 else []) + ([log_m] if (log_m <= 20 and args.cpu_seconds >= 10) else [min(18, log_m)])))
        e2e_runs = [cpu_end_to_end(lg) for lg in e2e_sizes]
        at_size = next((r_ for r_ in e2e_runs if r_["log2_constraints"] == log_m), None)
        out["cpu_baseline"] = {
            "value": (m / at_size["end_to_end_s"]) if at_size else m / cpu_s,
            "value_is": "measured: a real CPU Proof::prove at the bench size (end_to_end[-1])" if at_size else "composed from bounded samples (see sample)",
            "composed_value": m / cpu_s,
            "unit": "constraints/s",
            "cores": cores,
            "cpu_affinity_mask": mask,
            "cgroup_cpu_quota": quota,
            "kind": "port",
            "sample": ((f"one real CPU Proof::prove at 2^{log_m} on {cores} threads: {at_size['end_to_end_s']:.2f} s, its 118 bytes equal the GPU proof's; "
                        if at_size else "") +
                       f"composition cross-check: {n_s}-point per-point-scalar-mul MSM in {dt:.1f} s ({us_core:.1f} us*core/point) -> {msm_s_per_proof:.2f} s, "
                       f"extends {ext_s_per_proof:.2f} s, pointwise {pointwise_s_per_proof:.3f} s per proof"),
            "cpu_mhz": mhz,
            "msm_sample_points": n_s, "msm_sample_s": dt, "pippenger_sample_points": n_p,
            "end_to_end_s": e2e_runs[-1]["end_to_end_s"],
            "end_to_end": e2e_runs,
            "msm_points_per_s": pts_per_s,
            "msm_us_core_per_point": us_core,
            "extend_s_per_proof": ext_s_per_proof,
            "pointwise_s_per_proof": pointwise_s_per_proof,
            "best_cpu_pippenger_points_per_s": pip_pts_per_s,
            "best_cpu_pippenger_constraints_per_s": m / ((inst.n_wires + 5 * m) / pip_pts_per_s + ext_s_per_proof + pointwise_s_per_proof),
            "openssl_ec_point_mul_points_per_s": ossl,
        }
    detail_path = write_detail(out)
    line = json.dumps(headline(out, detail_path))
    assert len(line) < LINE_LIMIT, f"bench line is {len(line)} characters (limit {LINE_LIMIT}): move keys to the sidecar"
    print(line, flush=True)


if __name__ == "__main__":
    main()
