"""GPU: the N > 1 control flow on ONE device -- two ranks over gloo that both use cuda:0 (the 1-GPU box has no second GPU
for RCCL; the collective calls, their order and the sharding are the ones the 8-GPU run makes).

* bench.py --gpus 2 starts two ranks itself (a child ``torch.distributed.run``) and reports n_gpus 2;
* sum over the shards of a view-sharded step == the unsharded step ON THE HIP PATH, with the per-Gaussian pass split into
  Gaussian ranges whose all-reduces start behind each range (SyncFreeBatch.run_views(grad_chunks, on_chunk) +
  FlatGradients.all_reduce_rows), including a step in which one rank has to render a rejected view again.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

ROOT = util.ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.timeout(400)
def test_bench_starts_two_ranks(gpu_device, tmp_path):
    import signal
    import time
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONFAULTHANDLER="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # Its own session: if the ranks hang, the whole group gets SIGABRT (faulthandler prints every Python stack into the log) and the
    # assertion below shows where they stood.  A stall is a FAILURE, not something to retry: round 2 saw this pair stall about once in a
    # dozen full-suite runs and papered over it with a retry; round 3 ran the same two rehearsals 90 times in a row on the MI355X box, half
    # of them beside a third process holding a GPU context like the pytest parent does (tools/mr_loop.py, profiles/r03_multirank_loop.txt):
    # no stall, no failure -- the cause was not reproduced, so if it ever shows again the stacks must surface.
    out_f, err_f = open(str(tmp_path / "bench.out"), "w+"), open(str(tmp_path / "bench.err"), "w+")
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device", "--config", "2", "--steps", "2",
                          "--warmup", "2", "--views-per-gpu", "4", "--no-cpu"], stdout=out_f, stderr=err_f, text=True, env=env, cwd=ROOT, start_new_session=True)
    t0 = time.time()
    while p.poll() is None and time.time() - t0 < 150:
        time.sleep(0.2)
    hung = p.poll() is None
    if hung:
        os.killpg(p.pid, signal.SIGABRT)
        time.sleep(3)
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        p.wait()
    out_f.seek(0); err_f.seek(0)
    rc, stdout, stderr = p.returncode, out_f.read(), err_f.read()
    out_f.close(); err_f.close()
    assert not hung and rc == 0, ("stalled" if hung else f"rc {rc}", stderr[-6000:])
    line = [l for l in stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["frames_per_step"] == 8
    assert "world size 2" in out["config"]["parallelism"] and "Gaussian ranges" in out["config"]["parallelism"]
    assert out["value"] > 0 and out["config"]["frames_rerendered"] == 0


_WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["TGS_ROOT"])
from youreditableavatar_amd import scenes
from youreditableavatar_amd.multiview import FlatGradients, SyncFreeBatch, shard_views
from diff_gaussian_rasterization import GaussianRasterizationSettings, _C
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("TGS_BACKEND", "gloo")
alone = os.environ.get("TGS_EVEN_ALONE") == "1"          # world size 1 THROUGH the process group: the RCCL path on a one-GPU box
dev = torch.device("cuda:0")
if world > 1 or alone:
    import datetime
    torch.cuda.set_device(0)
    kw = dict(device_id=dev) if backend == "nccl" else {}
    dist.init_process_group(backend, timeout=datetime.timedelta(seconds=180), **kw)
P, W, H, D, V = 7000, 208, 144, 2, 6
LM = os.environ.get("TGS_LEVEL_MAJOR") == "1"            # SH gradients coefficient plane by coefficient plane (needs the stored degree 3: M = 16)
if LM:
    D = 3
cloud = scenes.make_cloud(P, D, seed=77, scale_mult=3.0)
ACT = os.environ.get("TGS_ACTIVE_DEGREE")               # render below the stored degree and reduce the live SH rows only
ACT = None if ACT is None else int(ACT)
if ACT is not None:
    D = ACT
t = lambda a, rg=False: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev).requires_grad_(rg)
names = ("means3D", "opacities", "scales", "rotations", "shs")
L = {n: t(cloud[n], True) for n in names}
flat = FlatGradients([L[n] for n in names], sh_params={4: 0}, level_major=LM)
cams = [scenes.orbit_camera(W, H, azimuth_deg=a) for a in np.linspace(0.0, 300.0, V)]
settings = [GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=t(c.bg), scale_modifier=1.0,
                                          viewmatrix=t(c.viewmatrix), projmatrix=t(c.projmatrix), sh_degree=D, campos=t(c.campos), prefiltered=False, debug=False)
            for c in cams]
dLs = [t(scenes.upstream_gradient(W, H, seed=300 + v)) for v in range(V)]
mine = list(shard_views(V, rank, world))
_C.set_deterministic(True)
batch = SyncFreeBatch(granule=256, streams=2)
pending, calls, n_handles = [], [], []
def on_chunk(first, count):
    calls.append((first, count))
    pending.extend(flat.all_reduce_rows(first, count, even_alone=alone, sh_degree=ACT))
res = {}
for step in range(4):
    if step == 3 and rank == world - 1:
        batch.bound = batch.bound // 3          # this rank has to render views again in this step; the other one does not
    calls.clear()
    batch.run_views([settings[v] for v in mine], L["means3D"], L["opacities"], L["shs"], L["scales"], L["rotations"], None, accumulate=False,
                    upstream_view=lambda i, img: dLs[mine[i]], grad_chunks=3, on_chunk=on_chunk)
    n_handles.append(len(pending))
    for w in pending:
        w.wait()
    pending.clear()
    torch.cuda.synchronize()
    res[f"flat{step}"] = flat.flat.cpu().numpy().copy()
    res[f"sh{step}"] = L["shs"].grad.contiguous().cpu().numpy()
    res[f"staged{step}"] = np.asarray(len(flat.__dict__.get("_stage", {})))
    res[f"calls{step}"] = np.asarray(calls)
res["rejected"] = np.asarray(batch.rejected)
res["handles"] = np.asarray(n_handles)
np.savez(os.environ["TGS_OUT"] + f".{rank}.npz", **res)
if world > 1 or alone:
    dist.barrier()
    dist.destroy_process_group()
"""


def _launch_workers(tmp_path, world, tag, extra_env=None):
    """Starts `world` worker processes (all on cuda:0) and returns what each saved.  Logs go to files (a rank blocked on a full pipe would
    stall its peer inside a collective), and the ranks are watched together: if one dies, the other is not left waiting in the rendezvous
    for the backend's timeout.  A stall is a FAILURE (faulthandler stacks of every rank in the message), not something to retry."""
    import signal
    import time
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    out = str(tmp_path / f"res_{tag}")
    port = _free_port()
    procs, logs = [], []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TGS_ROOT=ROOT, TGS_OUT=out, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONFAULTHANDLER="1", **(extra_env or {}))
        log = open(str(tmp_path / f"{tag}_r{rank}.log"), "w+")
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, cwd=ROOT, stdout=log, stderr=subprocess.STDOUT, text=True))
    deadline = time.time() + 150
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [r for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad or time.time() > deadline:
            failed = bad[0] if bad else -1
            for p in procs:
                if p.poll() is None:
                    p.send_signal(signal.SIGABRT)            # faulthandler: every thread's Python stack into the log
            time.sleep(3)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    tails = []
    for log in logs:
        log.seek(0); tails.append(log.read()[-3000:]); log.close()
    assert failed is None and all(p.wait() == 0 for p in procs), ("stalled" if failed == -1 else f"rank {failed} failed", tails)
    return [np.load(out + f".{r}.npz") for r in range(world)]


@pytest.mark.timeout(700)
def test_two_rank_step_on_the_hip_path_equals_unsharded(gpu_device, tmp_path):
    one = _launch_workers(tmp_path, 1, "w1")[0]
    two = _launch_workers(tmp_path, 2, "w2")
    for step in range(4):
        want = one[f"flat{step}"]
        assert np.abs(want).max() > 0
        for r in range(2):
            got = two[r][f"flat{step}"]
            # every rank holds the whole batch's gradient: the same numbers as the unsharded step up to the order of the fp32 sums
            # (the one-process run adds 6 views in registers, two ranks add 3 + 3 and then the two partial sums)
            assert util.rel_l2(got, want) <= 2e-6, (step, r)
        assert np.array_equal(two[0][f"flat{step}"], two[1][f"flat{step}"])
        # the same ranges in the same order on both ranks (collectives pair up), also when one of them had to render views again
        assert np.array_equal(two[0][f"calls{step}"], two[1][f"calls{step}"]) and len(two[0][f"calls{step}"]) == 3
    assert int(two[1]["rejected"]) >= 1 and int(two[0]["rejected"]) == 0


@pytest.mark.timeout(500)
def test_range_wise_reduce_through_rccl_at_world_size_one(gpu_device, tmp_path):
    """The production backend on the one GPU this pool has: backend "nccl" (= RCCL) at world size 1, with the early return of
    FlatGradients.all_reduce_rows bypassed -- RCCL initialisation with HSA_ENABLE_IPC_MODE_LEGACY=0, the coalesced all-reduce of five
    device slices per Gaussian range (one group launch), its ordering against the per-Gaussian pass on the step's stream and the waits at
    the end of the step all run for real.  A sum over one rank is the identity: the gradients must equal the run without a process group
    bit for bit, in all four steps (the last one re-renders views)."""
    plain = _launch_workers(tmp_path, 1, "plain")[0]
    rccl = _launch_workers(tmp_path, 1, "rccl", dict(TGS_BACKEND="nccl", TGS_EVEN_ALONE="1"))[0]
    assert list(plain["handles"]) == [0, 0, 0, 0]
    assert list(rccl["handles"]) == [3, 3, 3, 3]            # one coalesced collective per Gaussian range (not one per parameter slice)
    for step in range(4):
        assert np.array_equal(plain[f"flat{step}"], rccl[f"flat{step}"]), step
        assert np.array_equal(plain[f"calls{step}"], rccl[f"calls{step}"]) and len(rccl[f"calls{step}"]) == 3
    assert int(rccl["rejected"]) >= 1


@pytest.mark.timeout(700)
@pytest.mark.parametrize("active", [0, 1])
def test_live_sh_rows_reduce_through_rccl_and_two_ranks(gpu_device, tmp_path, active):
    """SH stored for degree 2, the step rendered at degree 0 / 1 (the reference's sh_levels schedule): FlatGradients.all_reduce_rows(sh_degree=)
    reduces the (D + 1)^2 live coefficients through staging slices.  Through RCCL at world size 1 the gradients equal the run without a
    process group bit for bit (pack, sum over one rank, unpack = identity; dead coefficients stay exactly zero); two ranks over gloo end
    with the unsharded step's gradients."""
    env = dict(TGS_ACTIVE_DEGREE=str(active))
    plain = _launch_workers(tmp_path, 1, "plain", env)[0]
    rccl = _launch_workers(tmp_path, 1, "rccl", dict(env, TGS_BACKEND="nccl", TGS_EVEN_ALONE="1"))[0]
    two = _launch_workers(tmp_path, 2, "w2", env)
    assert list(rccl["handles"]) == [3, 3, 3, 3]
    live = (active + 1) ** 2
    for step in range(4):
        want = plain[f"flat{step}"]
        sh = want[-7000 * 27:].reshape(7000, 9, 3)          # the SH parameter is the last one of the flat buffer
        assert np.abs(sh[:, :live]).max() > 0 and np.all(sh[:, live:] == 0)
        assert np.array_equal(want, rccl[f"flat{step}"]), step
        for r in range(2):
            assert util.rel_l2(two[r][f"flat{step}"], want) <= 2e-6, (step, r)
            assert np.all(two[r][f"flat{step}"][-7000 * 27:].reshape(7000, 9, 3)[:, live:] == 0)
        assert np.array_equal(two[0][f"flat{step}"], two[1][f"flat{step}"])


@pytest.mark.timeout(700)
@pytest.mark.parametrize("active", [0, 2])
def test_level_major_live_planes_reduce_without_staging(gpu_device, tmp_path, active):
    """Round 6: FlatGradients(level_major=True) -- SH stored for degree 3, the step rendered at degree 0 / 2.  The per-Gaussian pass writes dL_dsh
    plane by plane, all_reduce_rows hands the (D + 1)^2 leading planes of each range to the collective as slices of the flat buffer: nothing is
    staged (no pack in front of the collective, no unpack behind it), RCCL at world size 1 leaves the gradients bit for bit, two ranks over gloo
    end with the unsharded step's gradients, and the SH gradients equal the ROW-major run's bit for bit."""
    env = dict(TGS_ACTIVE_DEGREE=str(active), TGS_LEVEL_MAJOR="1")
    plain = _launch_workers(tmp_path, 1, "plain", env)[0]
    rccl = _launch_workers(tmp_path, 1, "rccl", dict(env, TGS_BACKEND="nccl", TGS_EVEN_ALONE="1"))[0]
    two = _launch_workers(tmp_path, 2, "w2", env)
    assert list(rccl["handles"]) == [3, 3, 3, 3]
    live = (active + 1) ** 2
    for step in range(4):
        want = plain[f"flat{step}"]
        sh = plain[f"sh{step}"]
        assert sh.shape == (7000, 16, 3) and np.abs(sh[:, :live]).max() > 0 and np.all(sh[:, live:] == 0)
        assert np.array_equal(want, rccl[f"flat{step}"]), step
        assert int(rccl[f"staged{step}"]) == 0 and int(two[0][f"staged{step}"]) == 0
        for r in range(2):
            assert util.rel_l2(two[r][f"flat{step}"], want) <= 2e-6, (step, r)
            assert np.all(two[r][f"sh{step}"][:, live:] == 0)
        assert np.array_equal(two[0][f"flat{step}"], two[1][f"flat{step}"])
