"""Debugging aid (GPU box): repeat the two N > 1 rehearsals of tests/test_gpu_multirank.py -- `bench.py --gpus 2 --backend gloo --same-device`
and the two-rank worker -- N times each, optionally beside a third process that holds a GPU context the way the pytest parent does in a
full-suite run.  A run that exceeds LIMIT seconds gets SIGABRT on every descendant (PYTHONFAULTHANDLER=1 -> Python stacks in the log) and is
counted as a stall.   python tools/mr_loop.py OUTDIR N_BENCH N_WORKER [--holder]"""
import os, signal, socket, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir, n_bench, n_worker = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
holder_on = "--holder" in sys.argv
os.makedirs(out_dir, exist_ok=True)
LIMIT = 90
env0 = dict(os.environ, PYTHONFAULTHANDLER="1", HSA_ENABLE_IPC_MODE_LEGACY="0", TGS_ROOT=ROOT)
for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
    env0.pop(k, None)


def descendants(pid):
    kids = {}
    for d in os.listdir("/proc"):
        if d.isdigit():
            try:
                st = open(f"/proc/{d}/stat").read().rsplit(")", 1)[1].split()
                kids.setdefault(int(st[1]), []).append(int(d))
            except OSError:
                pass
    tree, todo = [], [pid]
    while todo:
        x = todo.pop(); tree.append(x); todo += kids.get(x, [])
    return tree


def watch(procs, tag, log):
    """wait for all of procs; on a stall dump the trees, abort them; returns (stalled, seconds)"""
    t0 = time.time()
    while any(p.poll() is None for p in procs) and time.time() - t0 < LIMIT:
        if any(p.poll() not in (None, 0) for p in procs):
            break
        time.sleep(0.1)
    dt = time.time() - t0
    stalled = any(p.poll() is None for p in procs) and dt >= LIMIT
    if any(p.poll() is None for p in procs):
        for p in procs:
            if p.poll() is None:
                for x in descendants(p.pid):
                    try:
                        c = open(f"/proc/{x}/cmdline").read().replace("\0", " ")[:140]
                        st = open(f"/proc/{x}/stat").read().rsplit(")", 1)[1].split()[0]
                        log.write(f"{tag}: PID {x} state {st} wchan {open(f'/proc/{x}/wchan').read()} :: {c}\n")
                        for tid in os.listdir(f"/proc/{x}/task"):
                            try:
                                log.write(f"    tid {tid} wchan {open(f'/proc/{x}/task/{tid}/wchan').read()} comm {open(f'/proc/{x}/task/{tid}/comm').read().strip()}\n")
                            except OSError:
                                pass
                    except OSError:
                        pass
                log.flush()
                for x in reversed(descendants(p.pid)):
                    try:
                        os.kill(x, signal.SIGABRT)
                    except OSError:
                        pass
        time.sleep(3)
        for p in procs:
            if p.poll() is None:
                for x in reversed(descendants(p.pid)):
                    try:
                        os.kill(x, signal.SIGKILL)
                    except OSError:
                        pass
    for p in procs:
        p.wait()
    return stalled, dt


holder = None
if holder_on:
    code = ("import torch, time, sys; sys.path.insert(0, %r)\n"
            "x = torch.randn(1 << 28, device='cuda:0'); ss = [torch.cuda.Stream() for _ in range(6)]\n"
            "for s in ss:\n"
            "    with torch.cuda.stream(s): y = x * 2\n"
            "torch.cuda.synchronize(); print('holder ready', flush=True); time.sleep(100000)\n" % ROOT)
    holder = subprocess.Popen([sys.executable, "-c", code], env=env0, stdout=subprocess.PIPE, text=True)
    holder.stdout.readline()

src = open(os.path.join(ROOT, "tests", "test_gpu_multirank.py")).read()
worker = os.path.join(out_dir, "worker.py")
open(worker, "w").write(src.split('_WORKER = r"""')[1].split('"""')[0])
summary = open(os.path.join(out_dir, "summary.txt"), "a")
log = open(os.path.join(out_dir, "stalls.log"), "a")
stalls = {"bench": 0, "worker": 0}
fails = {"bench": 0, "worker": 0}
times = {"bench": [], "worker": []}
for i in range(n_bench):
    o, e = open(os.path.join(out_dir, f"bench_{i}.out"), "w"), open(os.path.join(out_dir, f"bench_{i}.err"), "w")
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device", "--config", "2", "--steps", "2",
                          "--warmup", "2", "--views-per-gpu", "4", "--no-cpu"], stdout=o, stderr=e, env=env0, cwd=ROOT, start_new_session=True)
    st, dt = watch([p], f"bench_{i}", log)
    o.close(); e.close()
    stalls["bench"] += st; fails["bench"] += (p.returncode != 0 and not st); times["bench"].append(dt)
    if not st and p.returncode == 0:
        os.remove(os.path.join(out_dir, f"bench_{i}.out")); os.remove(os.path.join(out_dir, f"bench_{i}.err"))
    print(f"bench {i}: stalled={st} rc={p.returncode} {dt:.1f}s", flush=True)
for i in range(n_worker):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    procs, files = [], []
    for r in range(2):
        f = open(os.path.join(out_dir, f"worker_{i}_r{r}.log"), "w"); files.append(f)
        env = dict(env0, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TGS_OUT=os.path.join(out_dir, "res"))
        procs.append(subprocess.Popen([sys.executable, worker], env=env, cwd=ROOT, stdout=f, stderr=subprocess.STDOUT, start_new_session=True))
    st, dt = watch(procs, f"worker_{i}", log)
    for f in files:
        f.close()
    bad = any(p.returncode != 0 for p in procs)
    stalls["worker"] += st; fails["worker"] += (bad and not st); times["worker"].append(dt)
    if not st and not bad:
        for r in range(2):
            os.remove(os.path.join(out_dir, f"worker_{i}_r{r}.log"))
    print(f"worker {i}: stalled={st} rcs={[p.returncode for p in procs]} {dt:.1f}s", flush=True)
if holder is not None:
    holder.kill(); holder.wait()
line = (f"holder={holder_on} bench runs {n_bench} stalls {stalls['bench']} other failures {fails['bench']} (max {max(times['bench'] or [0]):.1f}s); "
        f"worker runs {n_worker} stalls {stalls['worker']} other failures {fails['worker']} (max {max(times['worker'] or [0]):.1f}s)")
print(line)
summary.write(line + "\n")
