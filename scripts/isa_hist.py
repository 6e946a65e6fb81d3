#!/usr/bin/env python3
"""Instruction histogram of the loops of one kernel in the gfx950 ISA (hipcc -save-temps)."""
import collections, os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else "eval_kernelILb1ELb1"
unit = sys.argv[2] if len(sys.argv) > 2 else "abi_solve.hip"  # the translation unit that launches the kernel (abi_batched.hip, abi_frontend.hip)
d = tempfile.mkdtemp()
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-mllvm", "-amdgpu-kernarg-preload-count=8", "-c",
                       os.path.join(root, "camlasercalibratool_amd/csrc", unit), "-save-temps", "-o", "x.o"],
                      cwd=d, stderr=subprocess.DEVNULL)
s = open(os.path.join(d, [f for f in os.listdir(d) if f.endswith("gfx950.s")][0])).read()
names = re.findall(r"^(_Z\w*%s\w*):" % pat, s, flags=re.M)
for name in names:
    i = s.index(name + ":"); j = s.index(".Lfunc_end", i)
    lines = s[i:j].split("\n")
    labels = [n for n, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)] + [len(lines)]
    print("==", name, "lines", len(lines))
    for a, b in zip(labels[:-1], labels[1:]):
        c = collections.Counter()
        for l in lines[a:b]:
            l = l.strip()
            if not l or l.startswith((".", ";", "//")): continue
            c[l.split()[0]] += 1
        tot = sum(c.values())
        if tot >= 60:
            print(lines[a].split(";")[0].strip(), "instr", tot, "VALU", sum(v for k, v in c.items() if k.startswith("v_")),
                  "f64", sum(v for k, v in c.items() if "f64" in k), "vmem", sum(v for k, v in c.items() if "global_" in k),
                  "| top:", ", ".join(f"{k}:{v}" for k, v in c.most_common(8)))
    m = re.search(r"\.vgpr_count:\s+(\d+)", s[s.index(".name:           " + name):]) if (".name:           " + name) in s else None
    if m: print("vgpr_count", m.group(1))
