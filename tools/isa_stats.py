#!/usr/bin/env python3
"""Register use, scratch and static instruction counts of the kernels in libcaster.so.

  python tools/isa_stats.py [--lib cuda-pathtracer_amd/libcaster.so] [--match k_trace] [--dump DIR]

Unbundles the gfx950 code object from the shared library, reads the kernel
descriptors' notes (VGPRs, SGPRs, LDS, scratch = spills) and counts the
instructions of each kernel's disassembly by unit (VALU / SALU / VMEM / LDS /
other).  Works without a GPU."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(lib, out_dir):
    co = os.path.join(out_dir, "caster.co")
    # the fat binary sits in the .hip_fatbin section as a clang offload bundle
    fat = os.path.join(out_dir, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co],
                          stderr=subprocess.DEVNULL)
    return co


def kernel_notes(co):
    txt = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    out = {}
    cur = None

    def close(rec):
        if rec and "name" in rec:
            out[rec["name"]] = rec

    for line in txt.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count":   # the first key of a kernel's record
            close(cur)
            cur = {"agpr": int(v)}
            continue
        if cur is None:
            continue
        if k in ("vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size",
                 "vgpr_spill_count", "sgpr_spill_count", "max_flat_workgroup_size"):
            cur[k] = int(v)
        if k == "name" and "name" not in cur and v.startswith("_Z"):
            cur["name"] = v.strip("'\"")
        if k == "symbol":
            cur["name"] = v.strip("'\"")[:-3]
    close(cur)
    return out


def classify(op):
    if op.startswith(("v_",)):
        return "valu"
    if op.startswith(("s_load", "s_buffer_load", "s_store")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def disasm_counts(co, dump_dir=None):
    txt = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], text=True)
    counts = {}
    name = None
    buf = []
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if name and dump_dir:
                open(os.path.join(dump_dir, re.sub(r"\W+", "_", name)[:120] + ".s"), "w").write("\n".join(buf))
            name = m.group(1)
            counts[name] = {"valu": 0, "salu": 0, "smem": 0, "vmem": 0, "lds": 0, "other": 0, "fma": 0, "total": 0}
            buf = []
            continue
        if name is None:
            continue
        buf.append(line)
        t = line.strip().split()
        if not t:
            continue
        op = t[0]
        if op.endswith(":") or op.startswith("//"):
            continue
        c = counts[name]
        c[classify(op)] += 1
        c["total"] += 1
        if op.startswith(("v_fma_f32", "v_fmac_f32", "v_pk_fma_f32")):
            c["fma"] += 1
    if name and dump_dir:
        open(os.path.join(dump_dir, re.sub(r"\W+", "_", name)[:120] + ".s"), "w").write("\n".join(buf))
    return counts


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, p.stdout.splitlines()))


def main():
    ap = argparse.ArgumentParser()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ap.add_argument("--lib", default=os.path.join(root, "cuda-pathtracer_amd", "libcaster.so"))
    ap.add_argument("--match", default="")
    ap.add_argument("--dump", default=None, help="directory that receives one .s file per kernel")
    a = ap.parse_args()
    if a.dump:
        os.makedirs(a.dump, exist_ok=True)
    with tempfile.TemporaryDirectory() as d:
        co = code_object(a.lib, d)
        notes = kernel_notes(co)
        counts = disasm_counts(co, a.dump)
    names = demangle(list(counts))
    print("%-60s %5s %5s %6s %7s %6s %6s %5s %5s %5s %5s" %
          ("kernel", "vgpr", "spill", "lds", "scratch", "total", "valu", "fma", "salu", "vmem", "lds"))
    for sym, c in sorted(counts.items(), key=lambda kv: names[kv[0]]):
        nm = names[sym]
        if a.match and a.match not in nm:
            continue
        n = notes.get(sym, {})
        short = re.sub(r"\(.*$", "", nm.replace("caster::", "").replace("void ", ""))
        print("%-60s %5s %5s %6s %7s %6d %6d %5d %5d %5d %5d" %
              (short[:60], n.get("vgpr_count", "?"), n.get("vgpr_spill_count", "?"), n.get("group_segment_fixed_size", "?"),
               n.get("private_segment_fixed_size", "?"), c["total"], c["valu"], c["fma"], c["salu"], c["vmem"], c["lds"]))


if __name__ == "__main__":
    sys.exit(main())
