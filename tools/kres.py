#!/usr/bin/env python3
"""usage: tools/kres.py <libttemb_*.so | gfx950 .s file> [substring...] [--fail-on-scratch]
VGPRs / scratch / LDS of every ttemb:: kernel -- from the code objects embedded in a built library (the numbers the
loader sees: AMDGPU metadata notes) or from a --save-temps assembly.  occupancy = 512 // vgprs waves per SIMD.
--fail-on-scratch: exit 1 when a ttemb:: kernel has private_segment_fixed_size > 0 (a spilling kernel is a routing or a
launch-bounds mistake here: tests/test_host_logic.py runs this check on the shipped library)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def from_asm(path):
    name, info = None, {}
    for l in open(path):
        m = re.match(r'\s*\.amdhsa_kernel (\S+)', l)
        if m:
            name = m.group(1)
            info[name] = {}
        if name:
            for k, short in (('.amdhsa_next_free_vgpr', 'vgpr'), ('.amdhsa_private_segment_fixed_size', 'scratch'),
                             ('.amdhsa_group_segment_fixed_size', 'lds')):
                m = re.match(r'\s*' + re.escape(k) + r' (\d+)', l)
                if m:
                    info[name][short] = int(m.group(1))
        if '.end_amdhsa_kernel' in l:
            name = None
    return info


def from_library(path):
    """Kernel metadata of every gfx950 code object bundled into a host ELF (llvm-objdump --offloading extracts them next to
    its input, so the library is copied into a scratch directory first)."""
    info = {}
    with tempfile.TemporaryDirectory() as tmp:
        lib = os.path.join(tmp, os.path.basename(path))
        shutil.copy(path, lib)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", lib], check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "hipv4-amdgcn" not in f:
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(tmp, f)], check=True, capture_output=True,
                                   text=True).stdout
            cur = {}
            for l in notes.split("\n"):   # a kernel's keys come in alphabetical order: .name ... .vgpr_spill_count
                m = re.match(r'\s*-?\s*\.(name|private_segment_fixed_size|group_segment_fixed_size|vgpr_count|vgpr_spill_count|agpr_count):\s*(\S+)', l)
                if not m:
                    continue
                k, v = m.group(1), m.group(2)
                if k == "agpr_count":   # first key of a kernel entry
                    cur = {"agpr": int(v)}
                elif k == "name":
                    if v.startswith("_Z") or not v.isidentifier() or "kernel" in v:   # (argument entries have .name too)
                        cur["name"] = v
                elif k == "private_segment_fixed_size":
                    cur["scratch"] = int(v)
                elif k == "group_segment_fixed_size":
                    cur["lds"] = int(v)
                elif k == "vgpr_count":
                    cur["vgpr"] = int(v)
                elif k == "vgpr_spill_count":
                    cur["spill"] = int(v)
                    if "name" in cur:
                        info[cur["name"]] = cur
                    cur = {}
    return info


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    fail_on_scratch = "--fail-on-scratch" in sys.argv
    path, pats = args[0], args[1:]
    info = from_asm(path) if path.endswith(".s") else from_library(path)
    names = [n for n in info if not pats or any(p in n for p in pats)]
    dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
    bad = []
    for n, d in sorted(zip(names, dem), key=lambda x: x[1]):
        if "ttemb::" not in d and not pats:
            continue
        i = info[n]
        short = re.sub(r'^void ttemb::', '', d.split('(')[0])
        print(f"{short[:90]:90s} vgpr {i.get('vgpr', 0):4d} waves/SIMD {512 // max(i.get('vgpr', 1), 1):2d} scratch {i.get('scratch', 0):5d} lds {i.get('lds', 0)}")
        if i.get("scratch", 0) > 0 and "ttemb::" in d:
            bad.append((short, i["scratch"]))
    if fail_on_scratch and bad:
        print("kernels with scratch: " + ", ".join(f"{n} ({b} B/lane)" for n, b in bad), file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
