"""Development build of the library for A/B timing: python tools/build_dev.py <name> [-DMACRO ...] [-mllvm -flag ...] [--plain] [--cpl 5,3]
-> tools/dev/_ab/lib_<name>.so with the kernels of the given cells-per-lane counts only (default 3 and 5)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import __graft_entry__ as ge
name = sys.argv[1]
args = sys.argv[2:]
cpls = (3, 5)
if "--plain" in args:          # no per-unit scheduler settings (they are also dropped when -mllvm flags are given)
    args.remove("--plain")
if "--cpl" in args:
    i = args.index("--cpl")
    cpls = tuple(int(x) for x in args[i + 1].split(","))
    del args[i:i + 2]
out = os.path.join(R, "tools", "dev", "_ab")
os.makedirs(out, exist_ok=True)
defines = tuple(args) + (f"-DHC_CPL_MASK={sum(1 << n for n in cpls)}",)
lib = ge.build_library(os.path.join(out, f"lib_{name}.so"), cpls=cpls, defines=defines,
                       obj_dir=os.path.join(out, f"obj_{name}"), force=True,
                       unit_flags="-mllvm" not in args and "--plain" not in sys.argv)
print(lib)
