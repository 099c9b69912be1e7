"""Registers, LDS and scratch of every kernel in libagx.so AS THE CODE OBJECTS STATE THEM (the .amdhsa metadata notes):
the library's offload bundles are extracted into a scratch directory (llvm-objdump --offloading), llvm-readelf --notes
lists every kernel's .vgpr_count / .agpr_count / .sgpr_count / .group_segment_fixed_size / .private_segment_fixed_size.
rocprofv3's VGPR_Count column is a granulated figure (it showed 100 for a kernel of 193 registers): summaries quote this
table instead.   usage: python tools/kernel_resources.py [--md] [path/to/libagx.so]"""
import os, re, shutil, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    try:
        tool = os.path.join(LLVM, "llvm-cxxfilt")
        tool = tool if os.path.exists(tool) else "c++filt"
        out = subprocess.run([tool], input="\n".join(names) + "\n", capture_output=True, text=True, check=True).stdout.splitlines()
        return [o.replace("(anonymous namespace)::", "").replace("void ", "") for o in out]
    except Exception:
        return names


def kernel_resources(lib=None):
    """-> {demangled kernel name (up to its argument list): dict(vgpr, agpr, sgpr, lds, scratch, wg)}"""
    lib = lib or os.path.join(ROOT, "accelerating-genomics_amd", "libagx.so")
    res = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        shutil.copy(lib, os.path.join(d, "lib.so"))
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, capture_output=True, check=True)
        for f in sorted(os.listdir(d)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(d, f)], capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
                blk = ".agpr_count:" + blk
                get = lambda key: (re.search(r"\.%s:\s+(\S+)" % key, blk) or [None, None])[1]
                name = get("name")
                if not name:
                    continue
                res[name] = dict(vgpr=int(get("vgpr_count") or 0), agpr=int(get("agpr_count") or 0), sgpr=int(get("sgpr_count") or 0),
                                 lds=int(get("group_segment_fixed_size") or 0), scratch=int(get("private_segment_fixed_size") or 0),
                                 wg=int(get("max_flat_workgroup_size") or 0))
    names = list(res)
    return {dm.split("(")[0]: res[n] for n, dm in zip(names, demangle(names))}


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    table = kernel_resources(args[0] if args else None)
    print("| kernel | VGPR | AGPR | SGPR | LDS B (static) | scratch B | max workgroup |\n|---|---|---|---|---|---|---|")
    for k in sorted(table):
        r = table[k]
        print("| `%s` | %d | %d | %d | %d | %d | %d |" % (k, r["vgpr"], r["agpr"], r["sgpr"], r["lds"], r["scratch"], r["wg"]))
