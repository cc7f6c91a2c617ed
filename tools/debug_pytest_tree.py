"""Debugging aid: run a pytest selection as a child; if it is still running after LIMIT seconds, record the process tree (state, wchan,
command line) and send SIGABRT to every descendant (exact PIDs from the PPID chain; PYTHONFAULTHANDLER=1 makes them print their stacks)."""
import os, signal, subprocess, sys, time
limit = int(sys.argv[1]); args = sys.argv[2:]
env = dict(os.environ, PYTHONFAULTHANDLER="1")
p = subprocess.Popen([sys.executable, "-m", "pytest", *args], env=env)
t0 = time.time()
while p.poll() is None and time.time() - t0 < limit:
    time.sleep(1)
if p.poll() is None:
    def children(pid):
        out = []
        for d in os.listdir("/proc"):
            if d.isdigit():
                try:
                    st = open(f"/proc/{d}/stat").read().rsplit(")", 1)[1].split()
                    if int(st[1]) == pid:
                        out.append(int(d))
                except OSError:
                    pass
        return out
    tree, todo = [], [p.pid]
    while todo:
        x = todo.pop(); tree.append(x); todo += children(x)
    for x in tree:
        try:
            cmd = open(f"/proc/{x}/cmdline").read().replace("\0", " ")[:200]
            st = open(f"/proc/{x}/stat").read().rsplit(")", 1)[1].split()[0]
            wchan = open(f"/proc/{x}/wchan").read()
            print(f"PID {x} state {st} wchan {wchan} :: {cmd}", flush=True)
        except OSError as e:
            print(x, e)
    for x in reversed(tree[1:]):
        try:
            os.kill(x, signal.SIGABRT)
        except OSError:
            pass
    time.sleep(5)
    if p.poll() is None:
        os.kill(p.pid, signal.SIGABRT)
sys.exit(p.wait())
