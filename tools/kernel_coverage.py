"""Which kernels of libgdm_hip.so does the GPU test suite never launch?  Reads a rocprofv3 kernel_stats csv of a suite run (see
tools/kernel_coverage.sh) and the __global__ definitions in csrc/*.hip; prints per kernel the launch count (all template instances
together) and lists the ones with none.  Development aid: an entry point whose fallback branch launched NOTHING went unnoticed for
three rounds because no test reached it."""
import csv, glob, os, re, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
defined = {}
for f in sorted(glob.glob(os.path.join(root, "geometric_aware_dense_matching_amd", "csrc", "*.hip"))):
    src = open(f).read()
    for m in re.finditer(r"__global__[^;{]*?\bvoid\s+([A-Za-z_][A-Za-z0-9_]*)\s*\(", src, re.S):
        defined.setdefault(m.group(1), os.path.basename(f))
counts = dict.fromkeys(defined, 0)
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        name = row.get("Name") or row.get("Kernel_Name") or ""
        calls = int(float(row.get("Calls") or row.get("Count") or 0))
        base = re.sub(r"\(.*", "", name.replace("void ", "").replace("(anonymous namespace)::", ""))
        base = re.sub(r"<.*", "", base).strip()
        if base in counts:
            counts[base] += calls
# instance level: the library's kernel symbols (host-side launch stubs, demangled) against the launched names
import subprocess
so = os.path.join(root, "geometric_aware_dense_matching_amd", "libgdm_hip.so")
syms = [l.split()[-1] for l in subprocess.run(["nm", so], capture_output=True, text=True).stdout.splitlines() if "__device_stub__" in l]
dem = subprocess.run(["c++filt"], input="\n".join(sorted(set(syms))), capture_output=True, text=True).stdout.splitlines()


def norm(n):
    n = n.replace("__device_stub__", "").replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n).strip()


instances = sorted(set(norm(d) for d in dem))
launched = set()
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        launched.add(norm(row.get("Name") or ""))
inst_never = [i for i in instances if i not in launched]
print("%d template instances in the library, %d launched by the suite, %d never:" % (len(instances), len(instances) - len(inst_never), len(inst_never)))
for i in inst_never:
    print("  never launched: %s" % i)
print()
never = sorted(k for k, v in counts.items() if v == 0)
print("%d kernels defined, %d launched by the suite, %d never:" % (len(counts), len(counts) - len(never), len(never)))
for k in never:
    print("  never launched: %-44s %s" % (k, defined[k]))
print()
for k, v in sorted(counts.items(), key=lambda kv: kv[1]):
    if v:
        print("  %8d  %-44s %s" % (v, k, defined[k]))
