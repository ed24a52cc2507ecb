"""Which kernels does `-mllvm -structurizecfg-skip-uniform-regions=true` change?

That default-off LLVM switch is what gets the interpreter-style kernels (site_rate_kernel, locus_value_kernel,
locus_grad2_kernel) under their register budgets, and it once mis-merged two stores in an unrelated helper kernel that
happened to share a translation unit with them.  It is therefore confined to translation
units of their own, and this script proves what it touches: every flagged unit is compiled to gfx950 assembly with and
without the switch and the functions whose code differs are listed.  tests/test_host_logic.py asserts that they are exactly
the kernels the switch is meant for.  CPU only (hipcc cross-compiles).

usage: python tools/flag_containment.py            -> prints {unit: [changed kernels]}"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FLAG = ["-mllvm", "-structurizecfg-skip-uniform-regions=true"]


def _asm(src, flags, out):
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "--offload-device-only", "-S",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tapir_amd", "csrc"), "-o", out, src] + flags
    subprocess.check_call(cmd)
    return open(out).read()


def _functions(text):
    """{symbol: body} of every function in an AMDGPU assembly listing; comments, labels' numbering and debug lines are
    kept out of the comparison (a function's code is what counts)."""
    out = {}
    for m in re.finditer(r"^(\w+):\s*; @\1\n(.*?)^\.Lfunc_end\d+:", text, flags=re.M | re.S):
        body = "\n".join(l.split(";")[0].rstrip() for l in m.group(2).splitlines())
        body = "\n".join(l for l in body.splitlines() if l.strip() and not l.lstrip().startswith((".loc", ".file", ".cfi")))
        out[m.group(1)] = body
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.split("\n")[:len(names)] if p.returncode == 0 else list(names)


def changed_kernels(unit):
    src = os.path.join(ROOT, "tapir_amd", "csrc", unit)
    with tempfile.TemporaryDirectory() as tmp:
        with ThreadPoolExecutor(2) as ex:
            a = ex.submit(_asm, src, FLAG, os.path.join(tmp, "on.s"))
            b = ex.submit(_asm, src, [], os.path.join(tmp, "off.s"))
            on, off = _functions(a.result()), _functions(b.result())
    assert on.keys() == off.keys() and on, (unit, sorted(on.keys() ^ off.keys()))
    diff = sorted(k for k in on if on[k] != off[k])
    return demangle(diff), demangle(sorted(on))


def flagged_units():
    import __graft_entry__ as ge
    return [u["src"] for u in ge.HIP_UNITS if "-structurizecfg-skip-uniform-regions=true" in u["flags"]]


if __name__ == "__main__":
    for u in flagged_units():
        diff, allk = changed_kernels(u)
        print(u, "-- functions:", len(allk), "changed by the switch:", len(diff))
        for k in diff:
            print("   ", k[:140])
        for k in allk:
            if k not in diff:
                print("    (identical)", k[:140])
