"""The RCCL (backend "nccl") code paths of the sharded drivers.

1. World ONE on the one GPU of the box: the all-gather of host records through device tensors and the device-to-device
   exchange of Gram column blocks -- pins the calls, dtypes and pointer plumbing an N-rank run uses.
2. World TWO over RCCL when at least two devices are visible (skipped on the one-GPU box): `bench.py --gpus 2` as the
   driver's scaling run starts it -- the default partition (replicated X, contiguous k-chunks stitched, all-gather of the
   IC curve and the chunks' last models), its opt-in cooperative-prefill second figure, the fold-sharded CV workload and
   the Cox k-path -- so that an 8-GPU run is not RCCL's first run of this code.

Both start other processes on the device: collected after every parity test (tests/conftest.py)."""
import glob
import json
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _devices():
    import torch
    return torch.cuda.device_count()


def _bench_nccl(extra, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("BESSX_BENCH_ONE_DEVICE", None)
    with tempfile.TemporaryDirectory(prefix="bessx_bench_err_") as errdir:
        env["BESSX_BENCH_ERRDIR"] = errdir
        env["BESSX_BENCH_DETAIL_PATH"] = os.path.join(errdir, "detail.json")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "3000", "--p", "800", "--kmax", "30",
                              "--k-true", "10", "--steps", "2", "--warmup", "1", "--gpus", "2"] + extra, cwd=ROOT,
                             capture_output=True, text=True, timeout=timeout, env=env)
        errs = []
        for f in sorted(glob.glob(os.path.join(errdir, "*.err"))):
            with open(f) as fh:
                errs.append(fh.read()[-4000:])
        assert out.returncode == 0, "bench.py --gpus 2 %s (nccl) -> rc %d\n%s\n==== stderr (tail)\n%s" % (
            " ".join(extra), out.returncode, "\n".join(errs) or "(no per-rank error file)", out.stderr[-1500:])
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1 and len(lines[0]) < 6500, out.stdout[-2000:]
        json.loads(lines[0])  # (the compact line the driver parses; the full record of the run is in the side file)
        with open(os.path.join(errdir, "detail.json")) as fh:
            return json.load(fh)


def test_bench_two_ranks_over_rccl_when_two_devices_are_visible(gpu):
    if _devices() < 2:
        pytest.skip("one device visible: the N = 2 RCCL run needs two (the one-device rehearsals use gloo)")
    d = _bench_nccl(["--rebalance", "off", "--coop-variant", "--prefill", "0"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    rep = d["kpath_chunks_vs_single_chain"]
    assert rep["supports_equal_to_single_chain"] == rep["of"] == 30 and rep["prefill_columns"] == 0
    assert "Gram column blocks" not in d["config"]["collective"]
    d = _bench_nccl(["--prefill", "64", "--pilot", "12,64", "--no-cpu-baseline"])  # Gram blocks device to device
    rep = d["kpath_chunks_vs_single_chain"]
    assert rep["supports_equal_to_single_chain"] == rep["of"] == 30 and rep["prefill_columns"] == 64
    d = _bench_nccl(["--shard", "replica", "--no-cpu-baseline"])
    assert d["scaling"] == "weak" and d["ic_curves_gathered"] == 2
    d = _bench_nccl(["--workload", "lm-cv-gs", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["selected_k"] >= 1
    d = _bench_nccl(["--workload", "cox-seq"])
    rep = d["kpath_chunks_vs_single_chain"]
    assert rep["supports_equal_to_single_chain"] == rep["of"]


def test_rccl_collectives_of_the_sharded_paths_at_world_one(gpu):
    """The nccl (= RCCL) code path of bess_amd.dist on the one GPU there is: a process group of ONE rank with backend
    "nccl" -- the all-gather of host records through device tensors, and the device-to-device exchange of Gram column
    blocks (exported from / imported into the library's cache through the tensors' device pointers).  More ranks need
    more devices; this pins the calls, dtypes and pointer plumbing the N-rank run uses."""
    import os
    import subprocess
    import sys
    code = r"""
import os, sys
import numpy as np
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from bess_amd import capi, synth, dist as bdist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29713")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
comm = bdist._TorchComm("cuda")
got = comm.all_gather(np.arange(7.0), 1)
assert len(got) == 1 and np.array_equal(got[0], np.arange(7.0))
X, y, _, _ = synth.make_lm(1500, 400, 10)
seq = np.arange(1, 25)
with capi.Session(X, y, score_mode=2) as s:
    plain = s.sequential_path(seq, ic_type=3)
    cols = np.argsort(-s.marginal_scores(), kind="stable")[:96].astype(np.int32)
    s.cov_prefill_begin(cols)
    s.cov_prefill_compute(0, 3)
    want = s.cov_prefill_export(0, 3)
    comm.exchange_blocks(s, 1, 0, 3)            # device path: export into a cuda tensor, nccl all_gather
    buf = torch.empty(3 * 32 * 400, dtype=torch.float64, device="cuda")
    s.cov_prefill_export(0, 3, device_ptr=buf.data_ptr())
    assert np.array_equal(buf.cpu().numpy(), want)
    s.cov_prefill_import(1, 1, device_ptr=buf.data_ptr() + 32 * 400 * 8)   # a block back in through its device pointer
    assert np.array_equal(s.cov_prefill_export(0, 3), want)
    s.cov_prefill_end()
    rep = bdist.StitchedKPath(s, seq, 1, 0, ic_type=3, comm=comm).step()
assert np.array_equal(rep["chunk"]["cand_support"], plain["cand_support"])
np.testing.assert_allclose(rep["ic_curve"], plain["cand_ic"], rtol=1e-12)
dist.destroy_process_group()
print("NCCL-OK")
"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "NCCL-OK" in out.stdout, (out.stdout[-800:], out.stderr[-2500:])


def test_the_librarys_own_communicator_at_world_one(gpu):
    """bessx_comm_* (include/bessx.h section 5, round 6): RCCL behind the C ABI, loaded at run time -- a communicator of ONE
    rank on the one GPU there is: unique id, init, all-gather of host records through the library's device buffers, and
    the stitched k-path driven through it (what a C or R host has in place of torch.distributed).  In a process of its
    own WITHOUT torch: the library must find RCCL by itself."""
    import os
    import subprocess
    import sys
    code = r"""
import os, sys, faulthandler
faulthandler.dump_traceback_later(90, exit=True)
import numpy as np
sys.path.insert(0, %r)
from bess_amd import capi, synth, dist as bdist
assert "torch" not in sys.modules
ident = bdist.BessxComm.unique_id()
assert len(ident) == 128
comm = bdist.BessxComm(0, 1, ident, device=0)
got = comm.all_gather(np.arange(7.0), 1)
assert len(got) == 1 and np.array_equal(got[0], np.arange(7.0))
big = comm.all_gather(np.linspace(0.0, 1.0, 5000), 1)          # (the device buffers grow)
assert np.array_equal(big[0], np.linspace(0.0, 1.0, 5000))
X, y, _, _ = synth.make_lm(1500, 400, 10)
seq = np.arange(1, 25)
with capi.Session(X, y) as s:
    plain = s.sequential_path(seq, ic_type=3)
    rep = bdist.StitchedKPath(s, seq, 1, 0, ic_type=3, comm=comm, coarse_lead=True).step()
assert np.array_equal(rep["chunk"]["cand_support"], plain["cand_support"])
np.testing.assert_allclose(rep["ic_curve"], plain["cand_ic"], rtol=1e-12)
comm.close()
print("BESSX-COMM-OK")
"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))],
                         capture_output=True, text=True, timeout=200, env=env)
    assert out.returncode == 0 and "BESSX-COMM-OK" in out.stdout, (out.stdout[-800:], out.stderr[-2500:])


def test_bench_two_ranks_through_the_librarys_own_communicator_when_two_devices_are_visible(gpu):
    if _devices() < 2:
        pytest.skip("one device visible: two RCCL ranks need two")
    d = _bench_nccl(["--comm", "bessx", "--no-cpu-baseline"])
    rep = d["kpath_chunks_vs_single_chain"]
    assert rep["supports_equal_to_single_chain"] == rep["of"] == 30 and "bessx_comm" in d["config"]["communicator"]
