#!/usr/bin/env python3
"""Which kernels of the library changed?  tools/isa_diff.py before.s after.s [-v]

Both files are `hipcc --cuda-device-only -S` dumps of the library's device code (tools/isa_kernel.sh's command).  Per kernel:
the instruction text between its label and the kernel descriptor (comments, directives and blank lines stripped, local labels
anonymised) is compared; kernels present in both files are reported as identical / changed (instruction counts, VGPRs), the
others as added / removed.
"""
import hashlib
import re
import sys


def kernels(path):
    text = open(path, errors="replace").read()
    meta = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        v = re.search(r"\.amdhsa_next_free_vgpr (\d+)", m.group(2))
        meta[m.group(1)] = int(v.group(1)) if v else -1
    out, name, body = {}, None, []
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m and m.group(1) in meta:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        t = line.split(";")[0].strip()
        if t.startswith(".amdhsa_kernel") or t.startswith(".section") or t.startswith(".Lfunc_end"):
            out[name] = body
            name = None
        elif t.startswith(".L") and t.endswith(":"):
            body.append("L:")
        elif t and not t.startswith("."):
            body.append(re.sub(r"\.LBB\d+_\d+", "LBB", t))
    return out, meta


def short(n):
    m = re.match(r"_ZN3dsr(\d+)", n)
    if m:
        k = int(m.group(1))
        s = n[len(m.group(0)):]
        return s[:k] + ("<" + hashlib.md5(n.encode()).hexdigest()[:6] + ">" if len(s) > k + 1 else "")
    return n[:60]


def main():
    a, ma = kernels(sys.argv[1])
    b, mb = kernels(sys.argv[2])
    same = changed = 0
    # a kernel whose template / argument list changed has a new mangled name: pair it with a removed one of the same body
    gone = {k for k in a if k not in b}
    for k in sorted(set(b) - set(a)):
        twin = next((g for g in sorted(gone) if a[g] == b[k] and short(g).split("<")[0] == short(k).split("<")[0]), None)
        if twin:
            print(f"renamed   {short(twin)} -> {short(k)}  identical body ({len(b[k])} instr)")
            gone.discard(twin); a.pop(twin); a[k] = b[k]; ma[k] = mb[k]
    for k in sorted(set(a) | set(b)):
        if k not in b:
            print(f"removed   {short(k)}  ({len(a[k])} instr)")
        elif k not in a:
            print(f"added     {short(k)}  ({len(b[k])} instr, {mb[k]} vgprs)")
        elif a[k] == b[k]:
            same += 1
            if "-v" in sys.argv:
                print(f"identical {short(k)}  ({len(a[k])} instr)")
        else:
            changed += 1
            print(f"CHANGED   {short(k)}  {len(a[k])} -> {len(b[k])} instr, vgprs {ma[k]} -> {mb[k]}")
    print(f"{same} kernels byte-identical, {changed} changed")


if __name__ == "__main__":
    main()
