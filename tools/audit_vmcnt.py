"""ISA audit of the counted waits of the fragment-streamed edge kernels (csrc/b3d_estream.hpp, b3d_edge2.hpp).

Those kernels issue their weight stream as LDS-DMA from inline asm and wait for it with `s_waitcnt vmcnt(N)` where N counts
the ordinary vector-memory instructions hipcc emits between the last DMA piece and the rendezvous (FwdHooks / BwdHooks).
If the compiler ever merges or drops one of those loads / stores, N over-counts and the wait returns before the pieces
have landed: a silent race.  This tool disassembles the gfx950 code objects of the SHIPPED library and checks, for every
rendezvous (an `s_waitcnt vmcnt(N)` directly followed by `s_barrier`) of every `es::edge_*_kernel`:

    N  <=  number of vector-memory instructions between the last `global_load_lds_dwordx4` in front of it and the wait

(slack = that number - N: 0 means the table is exact, > 0 means stores are drained early -- a cost, not a bug).  It also
checks that these kernels use no scratch (a scratch reload behind an LDS-DMA is a vmcnt(0) drain).

    python tools/audit_vmcnt.py [path/to/libb3d_hip.so]        exit status 1 on any violation
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _find_objdump():
    """llvm-objdump of the ROCm toolchain: $ROCM_PATH / $HIP_PATH / /opt/rocm, then PATH."""
    for root in (os.environ.get("ROCM_PATH"), os.environ.get("HIP_PATH"), "/opt/rocm"):
        if root:
            for sub in ("lib/llvm/bin", "llvm/bin", "bin"):
                c = os.path.join(root, sub, "llvm-objdump")
                if os.path.exists(c):
                    return c
    return shutil.which("llvm-objdump")


OBJDUMP = _find_objdump()
VMEM = re.compile(r"^(global_load|global_store|global_atomic|buffer_load|buffer_store|buffer_atomic|flat_load|flat_store|flat_atomic|scratch_load|scratch_store)")
KERNELS = ("edge_fwd_kernel", "edge_bwd_kernel")


def code_objects(lib, tmp):
    so = os.path.join(tmp, "lib.so")
    shutil.copy(lib, so)
    subprocess.run([OBJDUMP, "--offloading", so], check=True, capture_output=True, cwd=tmp)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "gfx950" in f)


def functions(co):
    """{mangled name: [instruction text, ...]} of one code object."""
    txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    out, cur = {}, None
    for ln in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is not None and ln.startswith("\t"):
            ins = ln.strip().split("//")[0].strip()
            if ins:
                cur.append(ins)
    return out


def audit(name, ins):
    """[(index of the rendezvous, N, vmem since the last DMA, ok)], scratch instruction count"""
    res, since, seen_dma, scratch = [], 0, False, 0
    for i, s in enumerate(ins):
        if s.startswith("global_load_lds"):
            since, seen_dma = 0, True
            continue
        if s.startswith("scratch_"):
            scratch += 1
        if VMEM.match(s):
            since += 1
            continue
        m = re.match(r"s_waitcnt\s+vmcnt\((\d+)\)$", s)
        if m and i + 1 < len(ins) and ins[i + 1].startswith("s_barrier"):
            n = int(m.group(1))
            res.append((len(res), n, since, (not seen_dma) or n <= since))
    return res, scratch


def demangle(name):
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or (OBJDUMP and os.path.join(os.path.dirname(OBJDUMP), "llvm-cxxfilt"))
    if tool and os.path.exists(tool):
        return subprocess.run([tool, name], capture_output=True, text=True).stdout.strip() or name
    return name


def main():
    """A LINT of the shipped ISA, not a proof of race freedom: the scan is linear (loop back-edges are not followed) and counts
    instructions by mnemonic."""
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "batch3dmot_amd", "libb3d_hip.so")
    if OBJDUMP is None:
        print("audit_vmcnt: llvm-objdump not found (looked under $ROCM_PATH, $HIP_PATH, /opt/rocm and on PATH): the ISA audit cannot run")
        return 2
    if not os.path.exists(lib):
        print(f"audit_vmcnt: {lib} not found (build it first: make -C batch3dmot_amd/csrc)")
        return 2
    bad = found = 0
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(lib, tmp):
            for name, ins in functions(co).items():
                if not any(k in name for k in KERNELS) or "N3b3d2es" not in name:       # namespace b3d::es, mangled
                    continue
                dem = demangle(name).split("(")[0]
                dem = re.sub(r"b3d::MPDims<[^>]*>", "D", dem)
                res, scratch = audit(name, ins)
                dmas = sum(1 for s in ins if s.startswith("global_load_lds"))
                if not res or not dmas:
                    continue
                found += 1
                viol = [r for r in res if not r[3]]
                slack = sum(r[2] - r[1] for r in res[1:] if r[1] > 0)
                counted = [(r[0], r[1], r[2]) for r in res if r[1] > 0]
                print(f"{dem}: {len(res)} rendezvous, {dmas} LDS-DMA instructions, counted waits (index, N, vmem since DMA): {counted}, "
                      f"slack {slack}, scratch instructions {scratch}, violations {len(viol)}")
                for r in viol:
                    print(f"   VIOLATION: rendezvous {r[0]} waits vmcnt({r[1]}) with only {r[2]} vector-memory instructions behind the last DMA piece")
                bad += len(viol) + (1 if scratch else 0)
    if found < 3:
        print(f"expected the forward, backward and last-layer backward edge kernels, found {found}")
        return 1
    print("OK" if not bad else f"{bad} problem(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
