"""Diagnosis builds of the library that never touch the product sources (round 4, NOTES C.3): copies irr_amd/csrc to a scratch directory,
applies the named variant there and links irr_amd/lib_<tag>/libirr_hip.so (load with IRR_HIP_LIB=...).

    python tools/build_variant.py slp      conv_small.hip built WITH the vectorisers (the state the lane deviation was found in)
    python tools/build_variant.py slp128   ... and conv_smallco_dgrad4_kernel capped at 128 VGPRs (amdgpu_waves_per_eu(4))
    python tools/build_variant.py fat      built without the vectorisers (the product flags), but conv_smallco_dgrad4_kernel made to
                                           allocate 136 VGPRs (an empty asm that clobbers v135)
"""
import os, re, shutil, sys, tempfile

tag = sys.argv[1]
os.environ["IRR_BUILD_TAG"] = tag
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irr_amd import build

top = tempfile.mkdtemp(prefix="irr_" + tag + "_")          # sources include "../../include/irr_hip.h"
shutil.copytree(os.path.join(build.ROOT, "include"), os.path.join(top, "include"))
tmp = os.path.join(top, "irr_amd", "csrc")
os.makedirs(tmp)
for f in os.listdir(build.CSRC):
    shutil.copy(os.path.join(build.CSRC, f), tmp)
p = os.path.join(tmp, "conv_small.hip")
s = open(p).read()
head = "__global__ __launch_bounds__(256) void conv_smallco_dgrad4_kernel("
assert s.count(head) == 1
if tag in ("slp", "slp128"):
    build.EXTRA["conv_small.hip"] = build.EXTRA.get("conv_small.hip", []) + ["-fslp-vectorize", "-fvectorize"]
if tag == "slp128":
    s = s.replace(head, "__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void conv_smallco_dgrad4_kernel(")
if tag == "fat":
    body = "  constexpr int U = 4;\n  const long hw = (long)H * W;\n  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;"
    assert s.count(body) == 1
    s = s.replace(body, '  asm volatile("" ::: "v135");\n' + body)
open(p, "w").write(s)
build.CSRC = tmp
print(build.build())
