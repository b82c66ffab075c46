"""Are the fp32 MFMA kernels still the round-4 kernels?  Compiles densecap_amd/csrc/mfma_gemm.hip as it is and as it was at a
reference commit (default: the round-4 head) for gfx950, device code only, and compares every kernel of the old build with its
counterpart in the new one instruction for instruction (labels and comments stripped; the new build's extra template argument
-- BF3 = 0 -- is the only difference allowed in a name).  usage: python tools/check_fp32_isa.py [old_commit]   (CPU only, ~3 min)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OLD = sys.argv[1] if len(sys.argv) > 1 else "68dc326"


def kernels(path):
    txt = open(path).read()
    out = {}
    for m in re.finditer(r"^(_ZN12_GLOBAL__N_1\w+):[^\n]*\n", txt, flags=re.M):
        end = txt.index("s_endpgm", m.end())
        body = [l.split(";")[0].rstrip() for l in txt[m.end():end].splitlines()]
        out[m.group(1)] = [re.sub(r"\.LBB\d+_", ".LBB_", l) for l in body if l.strip() and not l.strip().startswith((".", "%"))]
    return out


def build(src_hip, common_h, out_s, td):
    d = os.path.join(td, os.path.basename(out_s) + "_d")
    os.makedirs(os.path.join(d, "csrc")); os.makedirs(os.path.join(d, "include"))
    open(os.path.join(d, "csrc", "m.hip"), "w").write(src_hip)
    open(os.path.join(d, "csrc", "common.h"), "w").write(common_h.replace("../../include/", "../include/"))
    for h in os.listdir(os.path.join(ROOT, "include")):
        open(os.path.join(d, "include", h), "w").write(open(os.path.join(ROOT, "include", h)).read())
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S",
                           os.path.join(d, "csrc", "m.hip"), "-o", out_s], stderr=subprocess.DEVNULL)


def show(commit, path):
    return subprocess.check_output(["git", "-C", ROOT, "show", "%s:%s" % (commit, path)]).decode()


with tempfile.TemporaryDirectory() as td:
    build(show(OLD, "densecap_amd/csrc/mfma_gemm.hip"), show(OLD, "densecap_amd/csrc/common.h"), os.path.join(td, "old.s"), td)
    build(open(os.path.join(ROOT, "densecap_amd/csrc/mfma_gemm.hip")).read(), open(os.path.join(ROOT, "densecap_amd/csrc/common.h")).read(),
          os.path.join(td, "new.s"), td)
    o, n = kernels(os.path.join(td, "old.s")), kernels(os.path.join(td, "new.s"))
same, bad = 0, []
for k, v in o.items():
    cand = [kk for kk in n if kk == k or kk == k.replace("EEEv8GemmDesc", "ELi0EEEv8GemmDesc")]
    if cand and n[cand[0]] == v:
        same += 1
    else:
        bad.append(k)
print("fp32 kernels of %s: %d; identical in the working tree: %d; new kernels in the tree: %d" % (OLD, len(o), same, len(n) - len(o)))
for k in bad:
    print("  DIFFERS:", k)
sys.exit(1 if bad else 0)
