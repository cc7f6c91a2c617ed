"""Debugging aid: run a command as a child; if it is still running after LIMIT seconds, record the process tree (state, wchan, command
line) and send SIGABRT to every descendant (exact PIDs from the PPID chain; PYTHONFAULTHANDLER=1 makes Python processes print their
stacks).  python tools/debug_cmd_tree.py LIMIT cmd args..."""
import os, signal, subprocess, sys, time
limit = int(sys.argv[1]); cmd = sys.argv[2:]
env = dict(os.environ, PYTHONFAULTHANDLER="1")
p = subprocess.Popen(cmd, env=env)
t0 = time.time()
while p.poll() is None and time.time() - t0 < limit:
    time.sleep(0.5)
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
    print("HUNG after", limit, "s; process tree:", flush=True)
    for x in tree:
        try:
            c = open(f"/proc/{x}/cmdline").read().replace("\0", " ")[:160]
            st = open(f"/proc/{x}/stat").read().rsplit(")", 1)[1].split()[0]
            print(f"PID {x} state {st} wchan {open(f'/proc/{x}/wchan').read()} :: {c}", flush=True)
        except OSError as e:
            print(x, e)
    for x in reversed(tree):
        try:
            os.kill(x, signal.SIGABRT)
        except OSError:
            pass
    time.sleep(3)
    for x in reversed(tree):
        try:
            os.kill(x, signal.SIGKILL)
        except OSError:
            pass
sys.exit(p.wait() if p.poll() is None else p.returncode)
