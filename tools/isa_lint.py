#!/usr/bin/env python3
"""Per-kernel ISA summary of the shipped library: instructions, scratch accesses, s_waitcnt vmcnt(0), 64-bit address adds, MFMAs, LDS-DMA,
global loads / stores -- the things that cost silently (a scratch reload behind an LDS-DMA is a vmcnt(0) drain; 64-bit per-lane addresses
eat register pairs).  python tools/isa_lint.py [lib] [min_instructions]"""
import os, re, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import audit_vmcnt as av

lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "batch3dmot_amd", "libb3d_hip.so")
mins = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rows = []
with tempfile.TemporaryDirectory() as tmp:
    for co in av.code_objects(lib, tmp):
        for name, ins in av.functions(co).items():
            if len(ins) < mins:
                continue
            c = lambda pat: sum(1 for i in ins if re.match(pat, i))
            rows.append((len(ins), c(r"scratch_"), c(r"s_waitcnt vmcnt\(0\)"), c(r"v_lshl_add_u64|v_add_co_u32.*\n?"), c(r"v_mfma"), c(r"global_load_lds"),
                         c(r"global_load_dword"), c(r"global_store"), c(r"s_barrier"), name))
print(f"{'instr':>7} {'scratch':>7} {'vmcnt0':>6} {'add64':>6} {'mfma':>6} {'dma':>5} {'gload':>6} {'gstore':>6} {'barr':>5}  kernel")
import subprocess
for r in sorted(rows, reverse=True):
    nm = subprocess.run(["c++filt", r[-1]], capture_output=True, text=True).stdout.strip() if os.path.exists("/usr/bin/c++filt") else r[-1]
    nm = nm.replace("b3d::", "").replace("(anonymous namespace)::", "")
    print(f"{r[0]:7d} {r[1]:7d} {r[2]:6d} {r[3]:6d} {r[4]:6d} {r[5]:5d} {r[6]:6d} {r[7]:6d} {r[8]:5d}  {nm[:110]}")
