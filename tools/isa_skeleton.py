#!/usr/bin/env python3
"""Memory skeleton of one kernel's ISA: loads, stores, waits, barriers and branches in program order -- what shows whether independent loads
are in flight together or each is waited for on its own.   python tools/isa_skeleton.py tgs_backward.hip k_preprocess_bwd_batch_split [max lines]"""
import os, re, subprocess, sys, tempfile
src, pat = sys.argv[1], sys.argv[2]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 400
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                       *os.environ.get("TGS_DEFINES", "").split(), os.path.join(root, "youreditableavatar_amd", "csrc", src), "-o", out], stderr=subprocess.DEVNULL)
s = open(out).read()
names = [m.group(1) for m in re.finditer(r"^(_Z\w+):", s, re.M) if pat in m.group(1)]
for n in names:
    dn = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    i = s.index(n + ":"); j = s.index(".end_amdhsa_kernel", i)
    body = s[i:j].split("\n")
    print(f"== {dn[:120]}  ({len(body)} lines)")
    keep = ("global_load", "global_store", "global_atomic", "s_waitcnt", "s_cbranch", "s_branch", ".LBB", "s_load", "s_barrier", "buffer_", "ds_write", "ds_read", "s_endpgm")
    k = 0
    for ln, l in enumerate(body):
        t = l.strip()
        if t.startswith(keep):
            print(f"{ln:5d}: {t[:100]}"); k += 1
            if k >= lim: break
