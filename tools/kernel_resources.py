#!/usr/bin/env python3
"""Per-kernel resources of the built library, read from its gfx950 code objects (no GPU needed):
    python tools/kernel_resources.py [zero-chain_amd/libzkamd.so] [substring of the kernel name]
prints  <mangled name>  scratch <bytes per lane>  vgpr <n>  lds <bytes per workgroup>.
tests/test_asm_routines.py asserts with it that the generated assembly loops' kernels use NO scratch memory."""
import os, re, subprocess, sys, tempfile

LLVM = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_resources(so_path):
    """{mangled kernel name: {"scratch": bytes per lane, "vgpr": n, "lds": bytes}} over every code object in the library"""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so_path], check=True)
        data = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)] + [len(data)]
        for k in range(len(starts) - 1):
            part, co = os.path.join(tmp, "part.bin"), os.path.join(tmp, "part.co")
            open(part, "wb").write(data[starts[k]:starts[k + 1]])
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + part, "--output=" + co], check=True)
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
            for blk in notes.split("- .agpr_count")[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                out[name] = {"scratch": int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1)),
                             "vgpr": int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)),
                             "lds": int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", blk).group(1))}
    return out


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "zero-chain_amd", "libzkamd.so")
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, r in sorted(kernel_resources(so).items()):
        if pat in name:
            print("%-90s scratch %5d  vgpr %3d  lds %6d" % (name[:90], r["scratch"], r["vgpr"], r["lds"]))
