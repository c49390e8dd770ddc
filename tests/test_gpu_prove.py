"""GPU parity tests (-m gpu): Proof::prove end to end through the C ABI.

  * BASELINE config #1 (toy R1CS of src/dvsnark_test.rs:131-180): every stage vector, the challenge,
    a0/b0 and both commitments are compared bit-exactly with the golden values frozen from the
    oracle, and the designated-verifier check (src/srs.rs:374-428) accepts.
  * a 2^8-row synthetic R1CS: same comparison against the oracle run live.
  * 2^14 rows: verify == true, tamper -> false, unsatisfied witness -> DVP_EUNSAT with the row index.
"""
import json
import os
import random

import numpy as np
import pytest

import pyref as o
import c_oracle as co
from util import to_limbs, from_limbs, np_to_pt

pytestmark = pytest.mark.gpu
VEC = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_vectors.json")))


def H(s):
    return int(s, 16)


def rows_of(inst):
    rows = []
    for i in range(inst.n_rows):
        row = []
        for mt in (inst.l, inst.r, inst.o):
            a, b = int(mt.row_ptr[i]), int(mt.row_ptr[i + 1])
            row.append([(int(mt.wire[k]), int(mt.coeff[k])) for k in range(a, b)])
        rows.append(tuple(row))
    return rows


def test_transcript_and_blake3(dvp):
    for n, h in VEC["blake3"].items():
        assert dvp.proving.blake3(bytes(i % 251 for i in range(int(n)))).hex() == h
    assert hex(dvp.proving.transcript_challenge(bytes(range(30)), o.TOY_PUBLIC)) == VEC["challenge_toy"]


def check_against_oracle(dvp, inst, pub, prv, trap, golden=None):
    pv = dvp.proving.Prover(inst)
    td = dvp.srs.Trapdoor(*trap)
    # setup scalars vs the oracle's brute-force setup
    tree = o.FFTree(pv.log_m + 1)
    st = o.setup_srs_scalars(tree, rows_of(inst), from_limbs(inst.coeffs), inst.num_public_inputs, trap)
    g_m, g_q, g_k = dvp.srs.srs_scalars(pv, inst, td)
    assert from_limbs(g_m) == st["g_m"] + [0] * (inst.n_wires - len(st["g_m"]))
    assert from_limbs(g_q) == st["g_q"]
    for j in range(3):
        assert from_limbs(g_k[j]) == st["g_k"][j]
    assert from_limbs(pv.debug("bar_wts")) == st["tables"]["bar_wts"]
    assert from_limbs(pv.debug("z_vals2inv")) == st["tables"]["z_vals2inv"]
    d, d2 = pv.domains()
    assert (from_limbs(d), from_limbs(d2)) == tuple(tree.both_domains())
    srs = dvp.srs.verifier_runs_setup(pv, inst, td)
    pv.set_srs(srs)
    proof = pv.prove(pub, prv)

    def alpha_fn(dl):
        return o.transcript_challenge(co.xsk233_encode(co.k233_mulgen(dl)), pub)

    pr = o.prove_scalars(tree, st, pub, prv, alpha_fn)
    for key in ("a", "b", "c", "i", "a2", "b2", "c2", "i2", "r2", "q2", "ka", "kb", "kr"):
        assert from_limbs(pv.debug(key)) == pr[key], key
    for key in ("alpha", "a0", "b0", "i0", "r0"):
        assert from_limbs(pv.debug(key))[0] == pr[key], key
    assert proof.commit_p == co.xsk233_encode(co.k233_mulgen(pr["dl_commit_p"]))
    assert proof.kzg_k == co.xsk233_encode(co.k233_mulgen(pr["dl_kzg"]))
    assert proof.a0_fr() == (pr["a0"], True) and proof.b0_fr() == (pr["b0"], True)
    assert o.verify_dl(trap, pub, pr["dl_commit_p"], pr["dl_kzg"], pr["a0"], pr["b0"], pr["alpha"])
    assert dvp.srs.verify(td, pub, proof)
    assert dvp.proving.Proof.from_bits(proof.to_bits()) == proof
    assert proof.to_bits() == o.proof_to_bits(proof.commit_p, proof.kzg_k, pr["a0"], pr["b0"])
    if golden:
        assert proof.commit_p.hex() == golden["commit_p"] and proof.kzg_k.hex() == golden["kzg_k"]
        assert hex(pr["alpha"]) == golden["alpha"]
    return pv, td, proof


def test_toy_r1cs_config1(dvp):
    """test_dvsnark_prover_over_toy_r1cs, src/dvsnark_test.rs:131-180, on the GPU path."""
    toy = VEC["toy"]
    inst = dvp.gnark_r1cs.R1CSInstance.from_rows(dvp.gnark_r1cs.TOY_ROWS, dvp.gnark_r1cs.TOY_COEFFS, 2)
    assert inst.num_constraints == 8 and inst.n_wires == 8
    trap = tuple(H(x) for x in toy["trapdoor"])
    pv, td, proof = check_against_oracle(dvp, inst, o.TOY_PUBLIC, o.TOY_PRIVATE, trap, golden=toy)
    # wrong public input must be rejected by the verifier
    assert not dvp.srs.verify(td, [25, 13], proof)
    # a witness that violates row 2 (2z*1 = t) is reported, not proved
    bad = list(o.TOY_PRIVATE)
    bad[3] += 1
    with pytest.raises(dvp.DvpError) as ei:
        pv.prove(o.TOY_PUBLIC, bad)
    assert ei.value.status == -3 and ei.value.index == 2


def test_synthetic_2_8_vs_oracle(dvp):
    inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(8)
    rnd = random.Random(8)
    trap = (rnd.randrange(1, o.P), rnd.randrange(1, o.P), rnd.randrange(1, o.P))
    check_against_oracle(dvp, inst, pub, prv, trap)


def test_synthetic_2_14_verifies(dvp):
    inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(14)
    rnd = random.Random(14)
    td = dvp.srs.Trapdoor(rnd.randrange(1, o.P), rnd.randrange(1, o.P), rnd.randrange(1, o.P))
    pv = dvp.proving.Prover(inst)
    pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
    proof = pv.prove(pub, prv)
    assert dvp.srs.verify(td, pub, proof)
    # proofs are deterministic (bit-identical bytes on a re-run)
    assert pv.prove(pub, prv) == proof
    tampered = dvp.proving.Proof(proof.commit_p, proof.kzg_k, (int.from_bytes(proof.a0, "little") ^ 1).to_bytes(29, "little"), proof.b0)
    assert not dvp.srs.verify(td, pub, tampered)
    assert not dvp.srs.verify(td, [pub[0], (pub[1] + 1) % o.P], proof)
    # fixed-base MSM mode (bases pre-rotated by tau^(c w), one shared 2^c-bucket set): same bytes for the
    # model-chosen window (single-level sort) and for forced c = 17 / 20 (two-level sort)
    for forced in (0, 17, 20):
        with dvp.tune(DVP_MSM_FIXED_MIN=1, DVP_MSM_FIXED_C=forced):
            pv3 = dvp.proving.Prover(inst)
            pv3.set_srs(dvp.srs.verifier_runs_setup(pv3, inst, td))
            assert pv3.prove(pub, prv) == proof, forced
            assert forced == 0 or pv3.msm_plan(1)[0] == forced
            pv3.close()
    # SRS handed over in the reference's file format (30-byte encodings) gives the same proof
    pv2 = dvp.proving.Prover(inst)
    srs = dvp.srs.verifier_runs_setup(pv2, inst, td)
    for which, (xy, inf) in enumerate(srs.as_list()):
        pv2.set_srs_encoded(which, dvp.curve.to_bytes(xy, inf))
    assert pv2.prove(pub, prv) == proof


def test_sparse_r1cs_via_dump_format(dvp):
    """BASELINE config #5 stand-in: an SP1-like sparse R1CS (1..8 terms per side, rows not a power of two,
    n_wires != m) serialised to the gnark dump format (src/gnark_r1cs.rs:84-91), parsed back, proved and
    verified; the unsatisfied-witness path reports the first bad row."""
    inst0, pub, prv = dvp.gnark_r1cs.synthetic_sparse(12)
    inst = dvp.gnark_r1cs.R1CSInstance.from_dump_bytes(inst0.to_dump_bytes(), 2)
    assert inst.n_rows == inst0.n_rows and inst.num_constraints == 4096 and inst.n_wires == inst0.n_wires
    rnd = random.Random(55)
    td = dvp.srs.Trapdoor(rnd.randrange(1, o.P), rnd.randrange(1, o.P), rnd.randrange(1, o.P))
    pv = dvp.proving.Prover(inst)
    pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
    proof = pv.prove(pub, prv)
    assert dvp.srs.verify(td, pub, proof)
    # evaluation vectors against the oracle's eval_row on a few rows, including a padded one
    a = from_limbs(pv.debug("a"))
    rows = rows_of(inst)
    w = [1] + pub + prv
    coeffs = from_limbs(inst.coeffs)
    for i in (0, 1, 77, inst.n_rows - 1):
        assert a[i] == o.eval_row(rows[i][0], coeffs, w)
    assert a[inst.n_rows] == 0 and a[-1] == 0
    bad = list(prv)
    bad[100] = (bad[100] + 1) % o.P
    with pytest.raises(dvp.DvpError) as ei:
        pv.prove(pub, bad)
    assert ei.value.status == -3 and 0 <= ei.value.index < inst.n_rows
    # witness file format (u32-BE count, 32-byte BE elements; src/gnark_r1cs.rs:58-77,188-198)
    blob = len(w).to_bytes(4, "big") + b"".join(x.to_bytes(32, "big") for x in w)
    assert dvp.gnark_r1cs.load_witness_bytes(blob) == w


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_prove_simulated_ranks(dvp, world):
    """The multi-GPU decomposition on ONE GPU: every 'rank' computes the partial MSM of its index range through
    dvp_prover_msm_partial, the partial points are combined exactly as GpuBackend.combine does after the
    all-gather, and the proof must be byte-identical to the single-GPU proof (only the RCCL call itself is
    not exercised here; tests/test_distributed_cpu.py covers the collective with gloo)."""
    import torch

    inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(13)
    rnd = random.Random(77)
    td = dvp.srs.Trapdoor(rnd.randrange(1, o.P), rnd.randrange(1, o.P), rnd.randrange(1, o.P))
    pv = dvp.proving.Prover(inst)
    pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
    ref = pv.prove(pub, prv)
    dev = torch.device("cuda", 0)
    be = dvp.distributed.GpuBackend(pv, dev)
    assignment = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).to(dev)
    # every simulated rank re-runs its own begin (with or without the extends, as shard_plan says) and, for the second
    # MSM, its own challenge.  Ranks that skip the extends run on a SECOND prover that never computed q2 / r2 / k_r, so
    # reading any of them would change the result.
    plan = dvp.distributed.shard_plan(world, *be.dims())
    pv2 = dvp.proving.Prover(inst)
    pv2.set_srs(dvp.srs.verifier_runs_setup(pv2, inst, td))
    be2 = dvp.distributed.GpuBackend(pv2, dev)
    parts = []
    for r in range(world):
        (lo, hi), _, need = plan[r]
        b = be if need else be2
        b.begin(assignment, need)
        parts.append(b.msm_partial(0, lo, hi).clone())
    commit = be.combine(torch.stack(parts))
    parts = []
    for r in range(world):
        _, (lo, hi), need = plan[r]
        b = be if need else be2
        b.begin(assignment, need)
        b.challenge(commit)
        parts.append(b.msm_partial(1, lo, hi).clone())
    be.begin(assignment, True)
    be.challenge(commit)
    proof = be.finish(be.combine(torch.stack(parts)))
    pv2.close()
    assert proof == ref and dvp.srs.verify(td, pub, proof)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_challenge_simulated_ranks(dvp, world):
    """The challenge phase sharded by index (dvp_prove_challenge_partial / _finish): every simulated rank
    inverts 1/(d - alpha) only on its slice of D and where its K-scalar range needs it, sums its slice of the three
    barycentric sums into a 128-byte record, the records are stacked as the all-gather would, and each rank forms only the K
    scalars of its own MSM range -- every rank on its own fresh prover, whose S, den and den2 hold nothing outside the
    rank's share, so a read outside it changes the result.  a0 b0 i0 r0 must equal the unsharded values on every rank and the proof must be
    byte-identical to the single-GPU proof."""
    import torch

    inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(12)
    rnd = random.Random(78)
    td = dvp.srs.Trapdoor(rnd.randrange(1, o.P), rnd.randrange(1, o.P), rnd.randrange(1, o.P))
    pv = dvp.proving.Prover(inst)
    pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
    ref = pv.prove(pub, prv)
    exp_abir0 = [from_limbs(pv.debug(n))[0] for n in ("a0", "b0", "i0", "r0")]
    dev = torch.device("cuda", 0)
    n_wires, m = inst.n_wires, pv.m
    plan = dvp.distributed.shard_plan(world, n_wires, m)
    assignment = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).to(dev)
    # one prover per simulated rank, as in a real run (each rank's den / S state is its own)
    provers = []
    for r in range(world):
        q = dvp.proving.Prover(inst)
        q.set_srs(dvp.srs.verifier_runs_setup(q, inst, td))
        provers.append(dvp.distributed.GpuBackend(q, dev))
    parts = []
    for r, be in enumerate(provers):
        (lo, hi), _, need = plan[r]
        be.begin(assignment, need)
        parts.append(be.msm_partial(0, lo, hi).clone())
    commit = provers[0].combine(torch.stack(parts))
    recs = []
    for r, be in enumerate(provers):
        rec = be.challenge_partial(commit, dvp.distributed.shard_range(m, r, world), plan[r][1])
        recs.append(rec.clone())
    gathered = torch.stack(recs)
    parts = []
    for r, be in enumerate(provers):
        be.challenge_finish(gathered[torch.randperm(world)], plan[r][1])      # the order of the records does not matter
        assert [from_limbs(be.prover.debug(n))[0] for n in ("a0", "b0", "i0", "r0")] == exp_abir0, r
        lo, hi = plan[r][1]
        parts.append(be.msm_partial(1, lo, hi).clone())
    proof = provers[0].finish(provers[0].combine(torch.stack(parts)))
    for be in provers:
        be.prover.close()
    assert proof == ref and dvp.srs.verify(td, pub, proof)


@pytest.mark.parametrize("horner_max_pub", [-1, 0])
def test_extends_by_vector_simulated_ranks(dvp, horner_max_pub):
    """The extends split by VECTOR over the ranks that need q2 / r2 (SURVEY 8e option A; distributed.prove_sharded from three
    extender ranks up): every simulated rank -- its own prover, begin without extends -- extends only the vectors it owns,
    the extended vectors are copied between the provers' own buffers through the zero-copy views the broadcast uses
    (GpuBackend.extended_tensor), and the quotient of every rank must equal the single-GPU prover's a2 b2 c2 r2 q2.  Both
    routes for i(X): Horner (3 vectors) and a fourth extend (DVP_HORNER_MAX_PUB = 0)."""
    import torch

    knobs = {} if horner_max_pub < 0 else {"DVP_HORNER_MAX_PUB": horner_max_pub}
    with dvp.tune(**knobs):
        inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(12)
        rnd = random.Random(79)
        td = dvp.srs.Trapdoor(rnd.randrange(1, o.P), rnd.randrange(1, o.P), rnd.randrange(1, o.P))
        pv = dvp.proving.Prover(inst)
        srs = dvp.srs.verifier_runs_setup(pv, inst, td)
        pv.set_srs(srs)
        ref = pv.prove(pub, prv)
        want = {n: pv.debug(n) for n in ("a2", "b2", "c2", "i2", "r2", "q2")}
        dev = torch.device("cuda", 0)
        assignment = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).to(dev)
        n_ext = pv.extend_count()
        assert n_ext == (3 if horner_max_pub < 0 else 4)
        ranks = []
        for r in range(3):
            q = dvp.proving.Prover(inst)
            q.set_srs(srs)
            ranks.append(dvp.distributed.GpuBackend(q, dev))
        ext_ranks = [0, 1, 2]
        for r, be in enumerate(ranks):
            be.begin(assignment, False)
            be.extend_vectors([v for v in range(n_ext) if dvp.distributed.extend_owner(v, ext_ranks) == r])
        for v in range(n_ext):  # the broadcast: the owner's vector into every other rank's buffer
            src = ranks[dvp.distributed.extend_owner(v, ext_ranks)].extended_tensor(v)
            for r, be in enumerate(ranks):
                if r != dvp.distributed.extend_owner(v, ext_ranks):
                    be.extended_tensor(v).copy_(src)
        torch.cuda.synchronize()
        # a vector that never arrived must stop the quotient (DVP_EINVAL), not give a silently wrong q2
        with pytest.raises(dvp.DvpError):
            ranks[0].quotient()
        for r, be in enumerate(ranks):
            for v in range(n_ext):
                if r != dvp.distributed.extend_owner(v, ext_ranks):
                    be.mark_extended(v)
        for be in ranks:
            be.quotient()
            for n in want:
                assert np.array_equal(be.prover.debug(n), want[n]), n
        # and the rest of the proof from one of them
        be = ranks[1]
        commit = be.msm_partial(0, 0, be.msm_size(0)).clone()
        be.challenge(commit)
        proof = be.finish(be.msm_partial(1, 0, be.msm_size(1)).clone())
        assert proof == ref
        for be in ranks:
            be.prover.close()
        pv.close()


def test_prove_2_20_full_size(dvp):
    """BASELINE config #4 at full size (2^20 constraints), oracle-backed (tests/fullsize.py): the domain and the SRS
    scalars are pinned on sampled indices by their definitions, commit_p / kzg_k by the discrete-log identity against
    the OpenSSL-pinned oracle (src/proving.rs:463,512,680), alpha by the oracle transcript, a0 / b0 by the barycentric
    formula in python big ints (src/ec_fft.rs:455-491), and the whole by the designated-verifier equation on discrete
    logs (src/srs.rs:374-428).  Then: reproducible bytes, tampering rejected, the sharded path (3 simulated ranks)
    and the in-library path over eight shards (the 8-GPU node's shape on one device) give identical bytes."""
    import torch
    import fullsize as fs

    rnd = random.Random(2020)
    inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(20)
    trap = (0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
    td = dvp.srs.Trapdoor(*trap)
    pv = dvp.proving.Prover(inst)
    scalars = dvp.srs.srs_scalars(pv, inst, td)
    g_m, g_q, g_k = scalars
    D, D2 = fs.check_domains(pv, 20, rnd)
    fs.check_srs_scalars(inst, trap, D, D2, g_m, g_q, g_k, rnd)
    mg = dvp.curve.point_scalar_mul_gen_batch
    pv.set_srs(dvp.srs.SRS(mg(g_m), mg(g_q), tuple(mg(v) for v in g_k)))
    proof = pv.prove(pub, prv)
    fs.check_proof(dvp, pv, inst, trap, pub, prv, proof, scalars, D=D, check_bary=True, rnd=rnd)
    assert dvp.srs.verify(td, pub, proof)
    assert pv.prove(pub, prv) == proof
    bad = dvp.proving.Proof(proof.commit_p, proof.kzg_k, proof.a0, (int.from_bytes(proof.b0, "little") ^ 2).to_bytes(29, "little"))
    assert not dvp.srs.verify(td, pub, bad)
    assert not dvp.srs.verify(td, [(pub[0] + 1) % o.P, pub[1]], proof)
    dev = torch.device("cuda", 0)
    assignment = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    pv.begin(assignment.data_ptr(), st)
    world = 3
    point = None
    for which in (0, 1):
        parts = []
        for r in range(world):
            lo, hi = dvp.distributed.shard_range(pv.msm_size(which), r, world)
            part = torch.zeros(10, dtype=torch.int64, device=dev)
            pv.msm_partial(which, lo, hi, part.data_ptr(), part.data_ptr() + 64, st)
            parts.append(part)
        be = dvp.distributed.GpuBackend(pv, dev)
        point = be.combine(torch.stack(parts))
        if which == 0:
            pv.challenge(point.data_ptr(), point.data_ptr() + 64, st)
    assert pv.finish(point.data_ptr(), point.data_ptr() + 64, st) == proof
    # the in-library multi-GPU path at full size with the shape of an 8-GPU node (dvp_set_devices([0] * 8): eight index-range shards
    # of both MSMs with their own tables, eight host threads, partial points added on the home device) -- same bytes, both flavours
    # of the transcript untouched by the device list
    try:
        dvp.set_devices([0] * 8)
        assert pv.prove(pub, prv) == proof
        assert pv.prove_dev(assignment.data_ptr(), st) == proof
    finally:
        dvp.set_devices([])
    assert pv.prove_dev(assignment.data_ptr(), st) == proof
    pv.close()


@pytest.mark.parametrize("log_m,devices", [(13, [0, 0]), (13, [0, 0, 0, 0, 0, 0, 0, 0]), (16, [0, 0, 0])])
def test_in_library_multi_gpu_same_bytes(dvp, log_m, devices):
    """dvp_set_devices: behind the unchanged dvp_prove signature the two MSMs are sharded over the listed
    devices -- per-device base slices and fixed-base tables, one host thread per device, partial points added on the home
    device.  Listing device 0 N times exercises the whole path on a one-GPU box; the bytes must not depend on the list,
    unsatisfied witnesses are still reported with their row, and a bad scalar index comes back in whole-vector terms."""
    inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(log_m)
    rnd = random.Random(100 + log_m)
    td = dvp.srs.Trapdoor(rnd.randrange(1, o.P), rnd.randrange(1, o.P), rnd.randrange(1, o.P))
    pv = dvp.proving.Prover(inst)
    pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
    with dvp.tune(DVP_MSM_FIXED_MIN=1):
        ref = pv.prove(pub, prv)
        assert dvp.srs.verify(td, pub, ref)
        try:
            dvp.set_devices(devices)
            assert pv.prove(pub, prv) == ref
            assert pv.prove(pub, prv) == ref          # shards are reused
            assert pv.msm_plan(1)[0] > 0
            bad = list(prv)
            bad[7] = (bad[7] + 1) % o.P
            with pytest.raises(dvp.DvpError) as e:
                pv.prove(pub, bad)
            assert e.value.status == -3
            dvp.set_devices(devices[:-1] if len(devices) > 2 else [])   # the list may change between proofs
            assert pv.prove(pub, prv) == ref
        finally:
            dvp.set_devices([])
        assert pv.prove(pub, prv) == ref
    with pytest.raises(dvp.DvpError):
        dvp.set_devices([0, 99])
    pv.close()


def test_table_flavours_same_bytes(dvp):
    """the two fixed-base table flavours -- the default aligned windows of signed binary digits (rows 2^(o_w) P) and round 2's
    aligned tau-adic windows (rows tau^(o_w) P, DVP_MSM_ALIGNED_SIGNED = 0): different recoders, table builders, bucket counts and
    tails -- produce the same proof; dvp_prover_msm_table_bytes reports flavour and size (rows x bases x 64 B)"""
    log_m = 13
    inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(log_m)
    rnd = random.Random(77)
    td = dvp.srs.Trapdoor(rnd.randrange(1, o.P), rnd.randrange(1, o.P), rnd.randrange(1, o.P))
    proofs = []
    for signed in (1, 0):
        with dvp.tune(DVP_MSM_FIXED_MIN=1, DVP_MSM_ALIGNED_SIGNED=signed):
            pv = dvp.proving.Prover(inst)
            pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
            assert pv.msm_table(0) == (0, False)      # no table before the first proof
            proofs.append(pv.prove(pub, prv))
            for which in (0, 1):
                nbytes, is_signed = pv.msm_table(which)
                assert nbytes == pv.msm_plan(which)[1] * pv.msm_size(which) * 64
                assert is_signed == bool(signed)
            pv.close()
    assert proofs[0] == proofs[1] and dvp.srs.verify(td, pub, proofs[0])


def test_points_sum_records(dvp):
    """dvp_points_sum_dev: n 80-byte records (x || y, u32 infinity flag) -> their sum; incl. the neutral element, P + P
    and P + (-P) (what a rank's all-gathered partial MSM results can be)"""
    import torch

    ks = [5, 7, 7, o.P - 7, 11]
    pts = [co.k233_mulgen(k) for k in ks]
    dev = torch.device("cuda", 0)
    rec = np.zeros((len(ks) + 1, 10), dtype=np.uint64)
    for i, pt in enumerate(pts):
        rec[i, :8] = to_limbs(list(pt)).reshape(8)
    rec[len(ks), 8] = 1  # a neutral partial
    d = torch.from_numpy(rec.view(np.int64)).to(dev)
    out = torch.zeros(10, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for n, exp_k in ((6, sum(ks) % o.P), (2, 12), (3, 19), (4, 12)):
        dvp.check(dvp.lib.dvp_points_sum_dev(d.data_ptr(), n, out.data_ptr(), out.data_ptr() + 64, st))
        h = out.cpu().numpy().view(np.uint64)
        got = None if (int(h[8]) & 0xFFFFFFFF) else tuple(from_limbs(h[:8].reshape(2, 4)))
        assert got == co.k233_mulgen(exp_k), n
    # two opposite points alone -> neutral
    d2 = torch.from_numpy(rec[[2, 3]].copy().view(np.int64)).to(dev)
    dvp.check(dvp.lib.dvp_points_sum_dev(d2.data_ptr(), 2, out.data_ptr(), out.data_ptr() + 64, st))
    assert int(out.cpu().numpy().view(np.uint64)[8]) & 0xFFFFFFFF == 1


@pytest.mark.parametrize("slots,devices", [(2, []), (1, []), (2, [0, 0])])
def test_two_provers_in_flight_same_bytes(dvp, slots, devices):
    """two and three provers on their own host threads and streams of ONE GPU (msm.hip: two MSM workspaces per device, the pair
    rounds of concurrent MSMs chained on the GPU by events -- HeavyGate; slots = 1: MSMs take turns): every proof of
    every thread equals the bytes a prover computes alone, for two circuit sizes at once (different plans, different
    workspace sizes, so a buffer shared by mistake would show) and with a stand-alone one-shot MSM running beside them.
    devices = [0, 0]: every prover additionally shards its MSMs over two in-library device entries (one host thread each), so
    up to six MSM shards contend for the two workspaces of device 0"""
    import threading

    import torch

    rnd = random.Random(4040)
    jobs = []
    for log_m in (13, 15, 13):
        inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(log_m)
        td = dvp.srs.Trapdoor(rnd.randrange(1, o.P), rnd.randrange(1, o.P), rnd.randrange(1, o.P))
        pv = dvp.proving.Prover(inst)
        pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
        w = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).cuda()
        jobs.append((pv, w, pv.prove_dev(w.data_ptr(), 0), torch.cuda.Stream()))
    # a one-shot MSM on a fourth thread: it takes whichever workspace is free (or waits for one)
    n = 5000
    k, s = to_limbs([rnd.randrange(o.P) for _ in range(n)]), to_limbs([rnd.randrange(o.P) for _ in range(n)])
    bases, _ = dvp.curve.point_scalar_mul_gen_batch(k)
    exp_msm = co.k233_mulgen(sum(a * b for a, b in zip(from_limbs(k), from_limbs(s))) % o.P)
    bad = []

    def prove_loop(i):
        pv, w, ref, st = jobs[i]
        for _ in range(6):
            if pv.prove_dev(w.data_ptr(), st.cuda_stream) != ref:
                bad.append(("proof", i))

    def msm_loop():
        for _ in range(6):
            xy, is_inf = dvp.curve.multi_scalar_mul(s, bases)
            if np_to_pt(xy, is_inf) != exp_msm:
                bad.append(("msm",))

    # a fifth thread keeps failing: an unsatisfied witness on a prover of its own must come back as DVP_EUNSAT with ITS row
    # index (the error index is thread-local) while the other threads' proofs stay right
    inst_u, pub_u, prv_u = dvp.gnark_r1cs.synthetic_dense(12)
    pv_u = dvp.proving.Prover(inst_u)
    pv_u.set_srs(dvp.srs.verifier_runs_setup(pv_u, inst_u, dvp.srs.Trapdoor(5, 6, 7)))
    prv_bad = list(prv_u)
    prv_bad[-1] = (prv_bad[-1] + 1) % o.P

    def unsat_loop():
        for _ in range(6):
            try:
                pv_u.prove(pub_u, prv_bad)
                bad.append(("unsat accepted",))
            except dvp.DvpError as e:
                if e.status != -3 or not (0 <= e.index < inst_u.n_rows):
                    bad.append(("unsat", e.status, e.index))

    with dvp.tune(DVP_MSM_WS_SLOTS=slots, DVP_MSM_FIXED_MIN=1, DVP_MSM_AFF_MIN=256):
        dvp.set_devices(devices)
        try:
            th = [threading.Thread(target=prove_loop, args=(i,)) for i in range(3)] + [threading.Thread(target=msm_loop), threading.Thread(target=unsat_loop)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        finally:
            dvp.set_devices([])
    torch.cuda.synchronize()
    assert not bad, bad
    for pv, *_ in jobs:
        pv.close()
    pv_u.close()


def _prof_count(dvp, name):
    import ctypes as C
    ms, n = C.c_double(0), C.c_uint64(0)
    dvp.check(dvp.lib.dvp_profile_read(name.encode(), C.byref(ms), C.byref(n)))
    return int(n.value)


def test_device_transcript_kernels_vs_oracle(dvp):
    """k_transcript / k_zalpha (what dvp_prove_dev runs between its MSMs since round 5) against the oracle's transcript
    (src/proving.rs:79-198) and the big-int vanishing polynomial, for 0 .. 35 public inputs (one BLAKE3 chunk of 29-byte inputs)"""
    inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(6)
    pv = dvp.proving.Prover(inst)
    rnd = random.Random(55)
    d, _ = pv.domains()
    dom = from_limbs(d)
    for npub in (0, 1, 2, 3, 17, 34, 35):
        pubs = [rnd.randrange(o.P) for _ in range(npub)]
        if npub >= 2:
            pubs[0], pubs[1] = 0, o.P - 1
        commit = bytes(rnd.randrange(256) for _ in range(30))
        alpha, neg_z = pv.transcript_dev(commit, pubs)
        assert alpha == o.transcript_challenge(commit, pubs), npub
        assert alpha == dvp.proving.transcript_challenge(commit, pubs), npub
        z = 1
        for x in dom:
            z = z * (alpha - x) % o.P
        assert neg_z == (-z) % o.P, npub
    assert hex(pv.transcript_dev(bytes(range(30)), o.TOY_PUBLIC)[0]) == VEC["challenge_toy"]
    with pytest.raises(dvp.DvpError):
        pv.transcript_dev(bytes(30), [1] * 36)
    pv.close()


@pytest.mark.parametrize("log_m", [8, 17])
def test_host_and_device_transcript_same_bytes_and_waits(dvp, log_m):
    """dvp_prove_dev with the transcript on the device (default) and on the host (rounds 1-4, DVP_PROVE_HOST_TRANSCRIPT=1): same 118
    bytes and same intermediates; the device flavour waits one time fewer on the proof's stream than the host flavour -- ONCE (the K
    MSM's final synchronisation) as soon as the MSMs are large enough for pair rounds (2^17), where the read of the largest bucket
    rides on a side stream behind the first round and is counted apart; a small MSM has to wait for that read mid-way"""
    import torch
    inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(log_m)
    rnd = random.Random(100 + log_m)
    td = dvp.srs.Trapdoor(rnd.randrange(1, o.P), rnd.randrange(1, o.P), rnd.randrange(1, o.P))
    pv = dvp.proving.Prover(inst)
    pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
    w = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for host in (1, 0, 1, 0):
        with dvp.tune(DVP_PROVE_HOST_TRANSCRIPT=host):
            pv.prove_dev(w.data_ptr(), st)  # tables, workspaces
            dvp.lib.dvp_profile_reset()
            proof = pv.prove_dev(w.data_ptr(), st)
            waits = (_prof_count(dvp, "host_waits_stream"), _prof_count(dvp, "host_waits_side"))
            mids = {k: from_limbs(pv.debug(k))[0] for k in ("alpha", "a0", "b0", "i0", "r0")}
        res.setdefault(host, (proof, mids, waits))
        assert res[host] == (proof, mids, waits)
        assert waits[0] + waits[1] == (4 if host else 3), (host, waits)
        if log_m >= 17:
            assert waits == ((2, 2) if host else (1, 2)), (host, waits)
    assert res[0][:2] == res[1][:2]
    assert dvp.srs.verify(td, pub, res[0][0])
    assert mids["alpha"] == o.transcript_challenge(res[0][0].commit_p, pub)
    pv.close()


def test_device_transcript_error_order(dvp):
    """the deferred flavour reports from ONE block at the end, in the reference's order: an unsatisfied row first
    (src/proving.rs:389-395), then a witness scalar >= p (the commitment MSM's range word, with its index)"""
    import torch
    inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(10)
    td = dvp.srs.Trapdoor(3, 5, 7)
    pv = dvp.proving.Prover(inst)
    pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
    st = torch.cuda.current_stream().cuda_stream
    good = dvp.fr.vec([1] + pub + prv)

    def dev(a):  # the device copy is bound to a name that outlives the prove_dev call reading it
        return torch.from_numpy(a.view(np.int64)).cuda()
    d_good = dev(good)
    proof = pv.prove_dev(d_good.data_ptr(), st)
    assert dvp.srs.verify(td, pub, proof)
    # (1) a wire no row reads correctly any more: unsatisfied row index
    bad = good.copy()
    bad[5, 0] ^= np.uint64(1)
    d_bad = dev(bad)
    with pytest.raises(dvp.DvpError) as ei:
        pv.prove_dev(d_bad.data_ptr(), st)
    assert ei.value.status == -3 and ei.value.index >= 0
    # (2) a non-canonical wire (value + p < 2^256): the rows are evaluated on whatever the limbs hold, so either a row is reported
    # (-3, as the reference asserts satisfaction first) or the commitment MSM's deferred range word names the wire (-1, index)
    k = 7
    v = int.from_bytes(good[k].tobytes(), "little") + o.P
    bad = good.copy()
    bad[k] = np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    d_bad2 = dev(bad)
    with pytest.raises(dvp.DvpError) as ei:
        pv.prove_dev(d_bad2.data_ptr(), st)
    assert ei.value.status == -3 or (ei.value.status, ei.value.index) == (-1, k)
    with dvp.tune(DVP_PROVE_HOST_TRANSCRIPT=1):
        with pytest.raises(dvp.DvpError) as ej:
            pv.prove_dev(d_bad2.data_ptr(), st)
    assert (ej.value.status, ej.value.index) == (ei.value.status, ei.value.index)
    # the prover is usable afterwards and gives the same bytes
    assert pv.prove_dev(d_good.data_ptr(), st) == proof
    pv.close()
