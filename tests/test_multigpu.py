"""Multi-GPU plumbing: bench.py's own N-rank launch, the C-ABI device selection, and the sharded
commit with the GPU engine under two ranks.  The two-device tests skip on a one-GPU box."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT, load_package

SETUP = os.path.join(ROOT, "tests", "golden", "trusted_setup.txt")


def test_bench_gpus_flag_launches_that_many_ranks():
    # no GPU here: every rank must start (RANK 0 and 1 both report) and fail loudly — not silently run one process
    if _gpu_count() > 0:
        pytest.skip("CPU-only check of the launcher")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, cwd=ROOT)
    out = p.stdout.decode()
    assert p.returncode != 0
    assert out.count("bench.py needs an MI355X") >= 2, out[-2000:]


def test_bench_rejects_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600, cwd=ROOT, env=env)
    assert p.returncode != 0
    assert "WORLD_SIZE=1" in p.stdout.decode()


def _gpu_count():
    try:
        return load_package().device_count()
    except Exception:
        return 0


@pytest.mark.gpu
def test_entry_points_leave_the_callers_device_alone(kzg):
    s = kzg.KZGSettings.from_file(SETUP)
    try:
        before = kzg.get_device()
        assert s.device() == before
        assert kzg.lib().kzgamd_msm_device(s.msm_handle()) == before
        blob = bytes(131072)
        assert kzg.blob_to_kzg_commitment(blob, s)[0] == 0xC0
        assert kzg.get_device() == before
    finally:
        s.close()


@pytest.mark.gpu
def test_reserve_then_enqueue_does_not_allocate(kzg):
    import torch

    s = kzg.KZGSettings.from_file(SETUP)
    try:
        dev = torch.device("cuda", 0)
        st = torch.cuda.Stream(device=dev)
        n = 8
        s.reserve(n, st.cuda_stream)
        free0 = torch.cuda.mem_get_info()[0]
        blobs = torch.zeros(n * 131072, dtype=torch.uint8, device=dev)
        out = torch.zeros(n * 48, dtype=torch.uint8, device=dev)
        stat = torch.zeros(n, dtype=torch.int32, device=dev)
        scratch = torch.empty(n * 131072, dtype=torch.uint8, device=dev)
        free1 = torch.cuda.mem_get_info()[0]
        kzg.blob_to_kzg_commitment_device(out.data_ptr(), stat.data_ptr(), scratch.data_ptr(), blobs.data_ptr(), n, s,
                                          st.cuda_stream)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free1  # the enqueue found its workspace already there
        assert free0 >= free1
        assert bytes(out.cpu().numpy().tobytes()[:48]) == bytes([0xC0]) + bytes(47)
    finally:
        s.close()


@pytest.mark.gpu
def test_two_devices_one_process(kzg, oracle, oracle_settings):
    import ctypes as C
    import random

    if kzg.device_count() < 2:
        pytest.skip("needs two GPUs")
    kzg.set_device(1)
    s1 = kzg.KZGSettings.from_file(SETUP)
    kzg.set_device(0)
    s0 = kzg.KZGSettings.from_file(SETUP)
    try:
        assert (s0.device(), s1.device()) == (0, 1)
        rnd = random.Random(11)
        blob = bytearray(rnd.randbytes(131072))
        for i in range(0, 131072, 32):
            blob[i] = 0
        blob = bytes(blob)
        want = C.create_string_buffer(48)
        assert oracle.lib().oblob_to_kzg_commitment(want, blob, C.byref(oracle_settings)) == 0
        # both from a thread whose current device is 0: the call on s1 switches to GPU 1 and back
        assert kzg.blob_to_kzg_commitment(blob, s1) == want.raw
        assert kzg.get_device() == 0
        assert kzg.blob_to_kzg_commitment(blob, s0) == want.raw
    finally:
        s0.close()
        s1.close()


GPU_WORKER = r'''
import ctypes as C, os, sys, random, importlib.util
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
from importlib import util
spec = util.spec_from_file_location("sharding", os.path.join(ROOT, "rust-kzg_amd", "sharding.py"))
sh = util.module_from_spec(spec); spec.loader.exec_module(sh)
torch.cuda.set_device(RANK)
path = os.path.join(ROOT, "rust-kzg_amd", "__init__.py")
spec = importlib.util.spec_from_file_location("rust_kzg_amd", path, submodule_search_locations=[os.path.dirname(path)])
kzg = importlib.util.module_from_spec(spec); sys.modules["rust_kzg_amd"] = kzg; spec.loader.exec_module(kzg)
import oracle_ffi as O
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % PORT, rank=RANK, world_size=2,
                        device_id=torch.device("cuda", RANK))
kzg.set_device(RANK)
s = kzg.KZGSettings.from_file(os.path.join(ROOT, "tests", "golden", "trusted_setup.txt"))
assert s.device() == RANK
rnd = random.Random(3)
blobs = []
for _ in range(9):
    b = bytearray(rnd.randbytes(131072))
    for i in range(0, 131072, 32):
        b[i] = 0
    blobs.append(bytes(b))
calls = []
def engine(bs):   # the GPU pipeline of this rank
    calls.append(len(bs))
    return kzg.blob_to_kzg_commitment_batch(b"".join(bs), len(bs), s)
got = sh.commit_sharded(blobs, engine, dist)
lo, hi = sh.shard_range(9, 2, RANK)
assert calls == [hi - lo], calls
with open(os.path.join(ROOT, "tests", "golden", "trusted_setup.txt"), "rb") as f:
    rc, os_ = O.load_settings(f.read())
for b, c in zip(blobs, got):
    o = C.create_string_buffer(48)
    assert O.lib().oblob_to_kzg_commitment(o, b, C.byref(os_)) == 0 and o.raw == c
dist.barrier()
dist.destroy_process_group()
s.close()
print("rank", RANK, "ok")
'''


ONE_GPU_WORKER = r'''
import ctypes as C, os, sys, random, importlib.util
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
from importlib import util
spec = util.spec_from_file_location("sharding", os.path.join(ROOT, "rust-kzg_amd", "sharding.py"))
sh = util.module_from_spec(spec); spec.loader.exec_module(sh)
torch.cuda.set_device(0)
path = os.path.join(ROOT, "rust-kzg_amd", "__init__.py")
spec = importlib.util.spec_from_file_location("rust_kzg_amd", path, submodule_search_locations=[os.path.dirname(path)])
kzg = importlib.util.module_from_spec(spec); sys.modules["rust_kzg_amd"] = kzg; spec.loader.exec_module(kzg)
import oracle_ffi as O
L = O.lib()
# two ranks, ONE GPU: RCCL refuses two ranks on one device, so the result gather runs over gloo on host tensors;
# everything that computes is this rank's own libkzg_mi355x.so context on GPU 0
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % PORT, rank=RANK, world_size=2)
kzg.set_device(0)
s = kzg.KZGSettings.from_file(os.path.join(ROOT, "tests", "golden", "trusted_setup.txt"))
assert s.device() == 0
with open(os.path.join(ROOT, "tests", "golden", "trusted_setup.txt"), "rb") as f:
    rc, os_ = O.load_settings(f.read())
assert rc == 0
rnd = random.Random(3)
blobs = []
for _ in range(9):
    b = bytearray(rnd.randbytes(131072))
    for i in range(0, 131072, 32):
        b[i] = 0
    blobs.append(bytes(b))
want = []
for b in blobs:
    o = C.create_string_buffer(48)
    assert L.oblob_to_kzg_commitment(o, b, C.byref(os_)) == 0
    want.append(o.raw)
# 1. commit_sharded: each rank commits to its slab with the GPU engine, results all-gathered as objects
calls = []
def engine(bs):
    calls.append(len(bs))
    return kzg.blob_to_kzg_commitment_batch(b"".join(bs), len(bs), s)
got = sh.commit_sharded(blobs, engine, dist)
lo, hi = sh.shard_range(9, 2, RANK)
assert calls == [hi - lo], calls
assert got == want
# 2. gather_results: the tensor form of the same gather (ncclAllGather on a multi-GPU node), device-resident proofs
proofs = kzg.compute_blob_kzg_proof_batch(b"".join(blobs[lo:hi]), b"".join(want[lo:hi]), hi - lo, s)
allp = sh.gather_results(torch.frombuffer(bytearray(b"".join(proofs)), dtype=torch.uint8), 9, 48, dist)
allp = bytes(allp.numpy().tobytes())
for i, b in enumerate(blobs):
    o = C.create_string_buffer(48)
    assert L.ocompute_blob_kzg_proof(o, b, want[i], C.byref(os_)) == 0
    assert allp[48 * i: 48 * i + 48] == o.raw, i
# 3. msm_sharded: one MSM split by index range, partial sums by the GPU engine, combined with kzgamd_g1_sum
n = 5000
g = O.G1(); L.og1_generator(C.byref(g))
pts = (O.G1Affine * n)()
base = O.G1(); k0 = O.fr_from_int(rnd.randrange(1, O.R)); L.og1_mul(C.byref(base), C.byref(g), C.byref(k0))
acc = O.G1(); C.memmove(C.byref(acc), C.byref(base), 144)
for i in range(n):   # P_i = base + i * G: cheap to generate, all distinct
    L.og1_to_affine(C.byref(pts[i]), C.byref(acc)); L.og1_add_or_dbl(C.byref(acc), C.byref(acc), C.byref(g))
sc = O.fr_array([rnd.randrange(O.R) for _ in range(n)])
def partial(lo, hi):
    out = kzg.multi_scalar_mult(C.cast(C.byref(pts, lo * 96), C.POINTER(kzg.BlstP1Affine)),
                                C.cast(C.byref(sc, lo * 32), C.POINTER(kzg.BlstFr)), hi - lo)
    return bytes(out)
total = sh.msm_sharded(n, partial, kzg.g1_sum, dist)
full = O.G1(); L.omsm_affine(C.byref(full), pts, sc, n)
a = O.G1(); C.memmove(C.byref(a), total, 144)
assert L.og1_equal(C.byref(a), C.byref(full)) == 1
dist.barrier()
dist.destroy_process_group()
s.close()
print("rank", RANK, "ok")
'''


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_with_the_gpu_engine(kzg):
    # SURVEY 8(e) on the hardware there is: both ranks open their own settings on GPU 0 (table budget capped so that
    # two fit), shard a commit batch, a proof batch and one MSM, gather over gloo, and every result equals the oracle's
    port = 31200 + (os.getpid() % 500)
    procs = []
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", KZGAMD_FBW_MAX_GB="40")
    for rank in range(2):
        code = "ROOT=%r\nPORT=%d\nRANK=%d\n" % (ROOT, port, rank) + ONE_GPU_WORKER
        procs.append(subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env))
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o[-3000:]
        assert "rank %d ok" % rank in o


RCCL_ONE_RANK_WORKER = r'''
import ctypes as C, os, sys, random, importlib.util
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
from importlib import util
spec = util.spec_from_file_location("sharding", os.path.join(ROOT, "rust-kzg_amd", "sharding.py"))
sh = util.module_from_spec(spec); spec.loader.exec_module(sh)
torch.cuda.set_device(0)
path = os.path.join(ROOT, "rust-kzg_amd", "__init__.py")
spec = importlib.util.spec_from_file_location("rust_kzg_amd", path, submodule_search_locations=[os.path.dirname(path)])
kzg = importlib.util.module_from_spec(spec); sys.modules["rust_kzg_amd"] = kzg; spec.loader.exec_module(kzg)
import oracle_ffi as O
L = O.lib()
# backend "nccl" IS RCCL on ROCm; one rank is what a one-GPU box can form (RCCL refuses two ranks on one device)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % PORT, rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
s = kzg.KZGSettings.from_file(os.path.join(ROOT, "tests", "golden", "trusted_setup.txt"),
                              kzg.make_config(table_budget_gb=40))
with open(os.path.join(ROOT, "tests", "golden", "trusted_setup.txt"), "rb") as f:
    rc, os_ = O.load_settings(f.read())
assert rc == 0
rnd = random.Random(3)
n = 9
raw = bytearray(rnd.randbytes(n * 131072))
for i in range(0, len(raw), 32):
    raw[i] = 0
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
d_blobs = torch.frombuffer(raw, dtype=torch.uint8).to(dev)
# this rank's slab through the device-resident pipeline: commitments and proofs never leave HBM before the collective
d_cm = torch.zeros(n * 48, dtype=torch.uint8, device=dev)
d_st = torch.zeros(n, dtype=torch.int32, device=dev)
d_cs = torch.empty(n * 131072, dtype=torch.uint8, device=dev)
kzg.blob_to_kzg_commitment_device(d_cm.data_ptr(), d_st.data_ptr(), d_cs.data_ptr(), d_blobs.data_ptr(), n, s, stream)
d_pr = torch.zeros(n * 48, dtype=torch.uint8, device=dev)
d_ps = torch.zeros(n, dtype=torch.int32, device=dev)
scr = torch.empty(n * kzg.PROOF_SCRATCH_BYTES, dtype=torch.uint8, device=dev)
kzg.compute_blob_kzg_proof_device(d_pr.data_ptr(), d_ps.data_ptr(), scr.data_ptr(), d_blobs.data_ptr(), d_cm.data_ptr(), n, s, stream)
torch.cuda.synchronize()
assert int(d_st.abs().sum()) == 0 and int(d_ps.abs().sum()) == 0
# the one collective of a sharded batch, on device tensors: ncclAllGather (all_gather_into_tensor) through RCCL
all_c = sh.gather_results(d_cm, n, 48, dist)
all_p = sh.gather_results(d_pr, n, 48, dist)
assert all_c.is_cuda and all_p.is_cuda
# and the timing collectives bench.py uses (barrier + MAX all_reduce over ranks)
t = torch.tensor([1.5], device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
assert float(t) == 1.5
cb, pb = all_c.cpu().numpy().tobytes(), all_p.cpu().numpy().tobytes()
for i in range(n):
    b = bytes(raw[i * 131072:(i + 1) * 131072])
    o = C.create_string_buffer(48)
    assert L.oblob_to_kzg_commitment(o, b, C.byref(os_)) == 0 and o.raw == cb[48 * i:48 * i + 48], i
    q = C.create_string_buffer(48)
    assert L.ocompute_blob_kzg_proof(q, b, o.raw, C.byref(os_)) == 0 and q.raw == pb[48 * i:48 * i + 48], i
dist.destroy_process_group()
s.close()
print("rccl single rank ok")
'''


@pytest.mark.gpu
def test_gather_results_runs_over_rccl_on_device_tensors(kzg):
    """the one collective of a sharded batch (sharding.gather_results -> all_gather_into_tensor ->
    ncclAllGather) and bench.py's barrier / MAX all-reduce EXECUTE through RCCL on device tensors — as a world of one
    rank, which is what a one-GPU box can form.  The slab comes from the device-resident commitment and proof pipelines,
    the gathered bytes are held to the oracle."""
    port = 32100 + (os.getpid() % 500)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    code = "ROOT=%r\nPORT=%d\n" % (ROOT, port) + RCCL_ONE_RANK_WORKER
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=900)
    o = p.stdout.decode()
    assert p.returncode == 0, o[-3000:]
    assert "rccl single rank ok" in o


@pytest.mark.gpu
def test_sharded_commit_gpu_engine_two_ranks(kzg):
    if kzg.device_count() < 2:
        pytest.skip("needs two GPUs")
    port = 30700 + (os.getpid() % 500)
    procs = []
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for rank in range(2):
        code = "ROOT=%r\nPORT=%d\nRANK=%d\n" % (ROOT, port, rank) + GPU_WORKER
        procs.append(subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env))
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % rank in o


@pytest.mark.gpu
def test_bench_two_ranks_on_a_two_gpu_box(kzg):
    import json

    if kzg.device_count() < 2:
        pytest.skip("needs two GPUs")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-extras", "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1800, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and sorted(d["device"] for d in line["devices"]) == [0, 1]


# ---------------------------------------------------------------- multi-GPU inside the library (csrc/multi.hip)
def _random_blobs(seed, n):
    import random

    rnd = random.Random(seed)
    blobs = []
    for _ in range(n):
        b = bytearray(rnd.randbytes(131072))
        for i in range(0, 131072, 32):
            b[i] = 0
        blobs.append(bytes(b))
    return blobs


def _in_process_multi(kzg, oracle, oracle_settings, devices, config=None):
    import ctypes as C
    import random

    L = oracle.lib()
    ms = kzg.MultiKZGSettings(SETUP, devices, config)
    try:
        assert ms.settings_devices() == list(devices)
        assert kzg.get_device() == 0  # loading on other devices left the caller's device alone
        blobs = _random_blobs(41, 37)
        want_c, want_p = [], []
        for b in blobs:
            o = C.create_string_buffer(48)
            assert L.oblob_to_kzg_commitment(o, b, C.byref(oracle_settings)) == 0
            want_c.append(o.raw)
            p = C.create_string_buffer(48)
            assert L.ocompute_blob_kzg_proof(p, b, o.raw, C.byref(oracle_settings)) == 0
            want_p.append(p.raw)
        # slabs of 19 + 18 (each object's pipelined large-batch path), 3 + 2 (lane path), 1 (first object only), 0
        for n in (37, 5, 1, 0):
            joined = b"".join(blobs[:n])
            assert ms.commit_batch(joined, n) == want_c[:n], n
            assert ms.proof_batch(joined, b"".join(want_c[:n]), n) == want_p[:n], n
        # page-locked caller buffers (kzgamd_pin_host_buffer: the copies of every slab are direct DMA): the same results
        pinned_in = C.create_string_buffer(b"".join(blobs), 37 * 131072)
        pinned_out = C.create_string_buffer(37 * 48)
        lib = kzg.lib()
        assert lib.kzgamd_pin_host_buffer(pinned_in, len(pinned_in)) == 0
        assert lib.kzgamd_pin_host_buffer(pinned_out, len(pinned_out)) == 0
        try:
            assert lib.kzgamd_blob_to_kzg_commitment_batch_multi(pinned_out, pinned_in, 37, ms.ptrs, ms.ndev) == 0
            assert pinned_out.raw == b"".join(want_c)
            C.memset(pinned_out, 0, len(pinned_out))
            assert lib.kzgamd_compute_blob_kzg_proof_batch_multi(pinned_out, pinned_in, b"".join(want_c), 37, ms.ptrs, ms.ndev) == 0
            assert pinned_out.raw == b"".join(want_p)
        finally:
            assert lib.kzgamd_unpin_host_buffer(pinned_in) == 0
            assert lib.kzgamd_unpin_host_buffer(pinned_out) == 0
        assert lib.kzgamd_unpin_host_buffer(pinned_out) == 1  # not registered any more
        assert lib.kzgamd_pin_host_buffer(None, 16) == 1
        # a blob with an element >= r in the SECOND slab fails the call like the reference (BadArgs)
        bad = bytearray(blobs[30])
        bad[64:96] = bytes.fromhex("73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001")
        spoiled = blobs[:30] + [bytes(bad)] + blobs[31:]
        with pytest.raises(kzg.KzgAmdError, match="C_KZG_RET 1"):
            ms.commit_batch(b"".join(spoiled), 37)
        with pytest.raises(kzg.KzgAmdError, match="C_KZG_RET 1"):
            ms.proof_batch(b"".join(spoiled), b"".join(want_c), 37)
        # batched verification: groups = slabs, verdicts ANDed (kzg/src/eip_4844.rs:770-816)
        allb, allc, allp = b"".join(blobs), b"".join(want_c), b"".join(want_p)
        assert ms.verify_blob_batch(allb, allc, allp, 37) is True
        swapped = want_p[:35] + [want_p[36], want_p[35]]
        assert ms.verify_blob_batch(allb, allc, b"".join(swapped), 37) is False
        assert ms.verify_blob_batch(b"", b"", b"", 0) is True
        # EIP-7594 cells + proofs, 3 blobs as 2 + 1, against the single-device entry point on the first object
        one = kzg.KZGSettings()  # a borrowed view of ms.arr[0] (loaded stays False: never freed through it)
        C.memmove(C.byref(one.c), C.byref(ms.arr[0]), C.sizeof(kzg.CKZGSettings))
        cells, proofs = ms.cells_and_proofs_batch(b"".join(blobs[:3]), 3)
        for i in range(3):
            c1, p1 = kzg.compute_cells_and_kzg_proofs(blobs[i], one)
            assert cells[i * 262144:(i + 1) * 262144] == c1
            assert proofs[i * 6144:(i + 1) * 6144] == p1
    finally:
        ms.close()
    # one MSM sharded by index range over two prepared handles: partials on the devices, sum on the host
    rnd = random.Random(5)
    n = 5000
    g = oracle.G1()
    L.og1_generator(C.byref(g))
    pts = (oracle.G1Affine * n)()
    base = oracle.G1()
    k0 = oracle.fr_from_int(rnd.randrange(1, oracle.R))
    L.og1_mul(C.byref(base), C.byref(g), C.byref(k0))
    acc = oracle.G1()
    C.memmove(C.byref(acc), C.byref(base), 144)
    for i in range(n):
        L.og1_to_affine(C.byref(pts[i]), C.byref(acc))
        L.og1_add_or_dbl(C.byref(acc), C.byref(acc), C.byref(g))
    sc = oracle.fr_array([rnd.randrange(oracle.R) for _ in range(n)])
    offsets = [0, 2600, n]
    handles = []
    try:
        for d in range(2):
            kzg.set_device(devices[d])
            handles.append(kzg.prepare_multi_scalar_mult(C.cast(C.byref(pts, offsets[d] * 96), C.POINTER(kzg.BlstP1Affine)),
                                                         offsets[d + 1] - offsets[d]))
        kzg.set_device(0)
        got = kzg.mult_pippenger_prepared_multi(handles, offsets, sc)
    finally:
        kzg.set_device(0)
        for h in handles:
            h.close()
    full = oracle.G1()
    L.omsm_affine(C.byref(full), pts, sc, n)
    a = oracle.G1()
    C.memmove(C.byref(a), bytes(got), 144)
    assert L.og1_equal(C.byref(a), C.byref(full)) == 1


@pytest.mark.gpu
def test_in_process_multi_two_settings_objects_on_gpu0(kzg, oracle, oracle_settings):
    # the in-library multi-GPU path on the hardware there is: two settings objects on GPU 0 (40 GB per table through
    # KzgAmdConfig.table_budget_bytes so that both fit — no environment variable), every result against the oracle
    _in_process_multi(kzg, oracle, oracle_settings, [0, 0], kzg.make_config(table_budget_gb=40))


_EIGHT_CACHE = {}


def _oracle_commitments_and_proofs(oracle, oracle_settings, blobs):
    """(commitments, proofs) of the oracle for `blobs`, computed once per session (both library flavours reuse them)."""
    import ctypes as C

    key = hash(blobs[0][:64]) ^ len(blobs)
    if key not in _EIGHT_CACHE:
        L = oracle.lib()
        cs, ps = [], []
        for b in blobs:
            o = C.create_string_buffer(48)
            assert L.oblob_to_kzg_commitment(o, b, C.byref(oracle_settings)) == 0
            p = C.create_string_buffer(48)
            assert L.ocompute_blob_kzg_proof(p, b, o.raw, C.byref(oracle_settings)) == 0
            cs.append(o.raw)
            ps.append(p.raw)
        _EIGHT_CACHE[key] = (cs, ps)
    return _EIGHT_CACHE[key]


@pytest.mark.gpu
def test_in_process_multi_eight_settings_objects_on_gpu0(kzg, oracle, oracle_settings):
    """BASELINE configs[4] in its exact shape — compute_blob_kzg_proof_batch over 256 blobs cut into EIGHT slabs of 32 —
    on the hardware there is: eight settings objects on GPU 0 (8 GB per table through KzgAmdConfig.table_budget_bytes),
    one host thread per object inside the library (csrc/multi.hip), the reference's batch parallelism
    (kzg/src/eip_4844.rs:770-816) with settings objects in the place of rayon workers.  Commitments, proofs, cells + cell
    proofs and batched verification of 256 blobs, then 257 (uneven slabs: 33 + 7 x 32), 7 and 3 blobs (fewer blobs than
    objects: empty slabs), every result against the oracle (cells: against ONE object's batch entry point, which the
    reference's cell vectors pin); one blob with an element >= r in slab 5 fails every call with C_KZG_BADARGS."""
    import ctypes as C

    NDEV, N = 8, 257
    blobs = _random_blobs(86, N)
    want_c, want_p = _oracle_commitments_and_proofs(oracle, oracle_settings, blobs)
    ms = kzg.MultiKZGSettings(SETUP, [0] * NDEV, kzg.make_config(table_budget_gb=8))
    try:
        assert ms.settings_devices() == [0] * NDEV
        infos = [ms.table_info(d) for d in range(NDEV)]
        assert all(i["rc"] == 0 and i["wide_table"] >= 1 for i in infos), infos  # every object got its own wide table (2: GLV form)
        one = kzg.KZGSettings()  # a borrowed view of the first object for the single-object reference calls
        C.memmove(C.byref(one.c), C.byref(ms.arr[0]), C.sizeof(kzg.CKZGSettings))
        for n in (256, 257, 7, 3):
            joined, cj, pj = b"".join(blobs[:n]), b"".join(want_c[:n]), b"".join(want_p[:n])
            assert ms.commit_batch(joined, n) == want_c[:n], n
            assert ms.proof_batch(joined, cj, n) == want_p[:n], n
            assert ms.verify_blob_batch(joined, cj, pj, n) is True, n
            cells, proofs = ms.cells_and_proofs_batch(joined, n)
            c1, p1 = kzg.compute_cells_and_kzg_proofs_batch(joined, n, one)
            assert cells == c1 and proofs == p1, n
            if n == 3:  # and the single-blob (direct) form, blob by blob
                for i in range(n):
                    cs, ps = kzg.compute_cells_and_kzg_proofs(blobs[i], one)
                    assert cells[i * 262144:(i + 1) * 262144] == cs and proofs[i * 6144:(i + 1) * 6144] == ps, i
        # the slabs of 256 blobs over 8 objects are 32 each: blob 170 lies in slab 5
        lo, hi = C.c_size_t(), C.c_size_t()
        assert kzg.lib().kzgamd_shard_range(256, NDEV, 5, C.byref(lo), C.byref(hi)) == 0
        assert (lo.value, hi.value) == (160, 192)
        bad = bytearray(blobs[170])
        bad[32 * 77:32 * 78] = bytes.fromhex("73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001")  # = r
        spoiled = b"".join(blobs[:170] + [bytes(bad)] + blobs[171:256])
        cj = b"".join(want_c[:256])
        with pytest.raises(kzg.KzgAmdError, match="C_KZG_RET 1"):
            ms.commit_batch(spoiled, 256)
        with pytest.raises(kzg.KzgAmdError, match="C_KZG_RET 1"):
            ms.proof_batch(spoiled, cj, 256)
        with pytest.raises(kzg.KzgAmdError, match="C_KZG_RET 1"):
            ms.cells_and_proofs_batch(spoiled, 256)
        with pytest.raises(kzg.KzgAmdError, match="C_KZG_RET 1"):
            ms.verify_blob_batch(spoiled, cj, b"".join(want_p[:256]), 256)
        # a wrong proof in slab 7 only: the ANDed verdict is false
        swapped = want_p[:254] + [want_p[255], want_p[254]]
        assert ms.verify_blob_batch(b"".join(blobs[:256]), cj, b"".join(swapped), 256) is False
        # the objects are still good after the failed calls
        assert ms.commit_batch(b"".join(blobs[:256]), 256) == want_c[:256]
    finally:
        ms.close()


@pytest.mark.gpu
def test_in_process_multi_two_gpus(kzg, oracle, oracle_settings):
    if kzg.device_count() < 2:
        pytest.skip("needs two GPUs")
    _in_process_multi(kzg, oracle, oracle_settings, [0, 1])


def test_multi_entry_points_reject_bad_arguments_without_a_gpu(kzg):
    import ctypes as C

    L = kzg.lib()
    out = C.create_string_buffer(48)
    blob = bytes(131072)
    empty = kzg.CKZGSettings()  # never loaded: not in the registry
    ptrs = (C.POINTER(kzg.CKZGSettings) * 1)(C.pointer(empty))
    assert L.kzgamd_blob_to_kzg_commitment_batch_multi(out, blob, 1, ptrs, 1) == kzg.C_KZG_BADARGS
    assert L.kzgamd_blob_to_kzg_commitment_batch_multi(out, blob, 1, ptrs, 0) == kzg.C_KZG_BADARGS
    assert L.kzgamd_blob_to_kzg_commitment_batch_multi(out, blob, 1, None, 1) == kzg.C_KZG_BADARGS
    assert L.kzgamd_compute_blob_kzg_proof_batch_multi(out, blob, out, 1, ptrs, 1) == kzg.C_KZG_BADARGS
    ok = C.c_bool(True)
    assert L.kzgamd_verify_blob_kzg_proof_batch_multi(C.byref(ok), blob, out, out, 1, ptrs, 1) == kzg.C_KZG_BADARGS
    arr = (kzg.CKZGSettings * 2)()
    assert L.kzgamd_load_trusted_setup_file_multi(arr, None, 2, None) == kzg.C_KZG_BADARGS
    L.kzgamd_free_trusted_setup_multi(arr, 2)  # empty objects: a no-op


def test_library_slabs_equal_the_python_partition(kzg):
    """kzgamd_shard_range (what the *_multi entry points cut a batch with) == sharding.shard_range, the partition of the
    torch.distributed form: both multi-GPU paths agree on who owns a blob.  No GPU needed."""
    import ctypes as C
    import importlib.util

    spec = importlib.util.spec_from_file_location("sharding", os.path.join(ROOT, "rust-kzg_amd", "sharding.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    L = kzg.lib()
    lo, hi = C.c_size_t(), C.c_size_t()
    for n in (0, 1, 5, 8, 9, 37, 256, 4097):
        for parts in (1, 2, 3, 8):
            prev = 0
            for k in range(parts):
                assert L.kzgamd_shard_range(n, parts, k, C.byref(lo), C.byref(hi)) == 0
                assert (lo.value, hi.value) == sh.shard_range(n, parts, k)
                assert lo.value == prev
                prev = hi.value
            assert prev == n
    assert L.kzgamd_shard_range(5, 0, 0, C.byref(lo), C.byref(hi)) == 1
    assert L.kzgamd_shard_range(5, 2, 2, C.byref(lo), C.byref(hi)) == 1
