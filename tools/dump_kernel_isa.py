#!/usr/bin/env python3
"""Disassembly of the shipped library's kernels whose (mangled) name contains PATTERN: python tools/dump_kernel_isa.py PATTERN [lib]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import audit_vmcnt as av

def main():
    pat = sys.argv[1]
    lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "batch3dmot_amd", "libb3d_hip.so")
    with tempfile.TemporaryDirectory() as tmp:
        for co in av.code_objects(lib, tmp):
            for name, ins in av.functions(co).items():
                if pat in name:
                    print("== %s (%d instructions)" % (name, len(ins)))
                    for i in ins:
                        print("   ", i)
main()
