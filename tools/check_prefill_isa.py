"""Static check of the prefill kernels' split LDS reads (csrc/mfma_attn_common.h: lds_issue_* + lds_wait_all).

Between an issue-only `ds_read_*` and the `s_waitcnt lgkmcnt(0)` that covers it the destination registers hold stale
data; the inline asm pins instruction ORDER, not register allocation, so a compiler-inserted copy or any other use of a
destination register inside that window would be silently wrong.  This script compiles paged_attn_prefill.hip to
assembly with the library's flags and fails if any instruction between an LDS read and the next full lgkm wait (in
program text order) names one of the read's destination registers.  Run by tests/test_build.py (CPU: hipcc only).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "vllm-omni_amd", "csrc", "paged_attn_prefill.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-pass-failed", "-fno-slp-vectorize", "--cuda-device-only", "-S"]


def regs(tok: str):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def split_kernels(asm: str):
    cur, body = None, []
    for raw in asm.split("\n"):
        line = raw.split(";")[0].strip()
        if not line:
            continue
        if line.endswith(":") and "prefill_mfma" in line and not line.startswith("."):
            cur, body = line[:-1], []
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):  # (an s_endpgm may sit in the middle of a kernel: early exits)
            yield cur, body
            cur = None
            continue
        if line.startswith(".") and not line.endswith(":"):
            continue                       # directives
        body.append(line)


def check_kernel(body):
    """Forward may-be-pending dataflow over the kernel's basic blocks (labels / branches from the text)."""
    blocks, names, cur = [], {}, []
    for line in body:
        if line.endswith(":"):
            if cur or not blocks:
                blocks.append(cur)
                cur = []
            names[line[:-1]] = len(blocks)
            continue
        cur.append(line)
        op = line.split()[0]
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm"):
            blocks.append(cur)
            cur = []
    blocks.append(cur)
    succ = []
    for i, blk in enumerate(blocks):
        out = set()
        last = blk[-1].split() if blk else [""]
        if last[0].startswith("s_cbranch"):
            out.add(names.get(last[1], None))
            out.add(i + 1)
        elif last[0] == "s_branch":
            out.add(names.get(last[1], None))
        elif last[0] != "s_endpgm":
            out.add(i + 1)
        succ.append({x for x in out if x is not None and x < len(blocks)})
    inn = [set() for _ in blocks]

    def transfer(i, report):
        pend = set(inn[i])
        for line in blocks[i]:
            op = line.split()[0]
            if op.startswith("s_waitcnt"):
                if "lgkmcnt(0)" in line:
                    pend = set()
                continue
            if op.startswith("ds_read"):
                dst = line.split()[1].rstrip(",")
                bad = regs(line[line.index(dst) + len(dst):]) & pend
                if bad and report is not None:
                    report.append((line, sorted(bad)))
                pend |= regs(dst)
                continue
            if pend:
                bad = regs(line) & pend
                if bad and report is not None:
                    report.append((line, sorted(bad)))
        return pend

    changed = True
    while changed:
        changed = False
        for i in range(len(blocks)):
            out = transfer(i, None)
            for j in succ[i]:
                if not out <= inn[j]:
                    inn[j] |= out
                    changed = True
    problems = []
    for i in range(len(blocks)):
        transfer(i, problems)
    return problems


def check(asm: str):
    kernels, problems = 0, []
    for name, body in split_kernels(asm):
        kernels += 1
        for line, bad in check_kernel(body):
            problems.append((name, 0, line, bad))
    return kernels, problems


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "prefill.s")
        subprocess.run([HIPCC, *FLAGS, SRC, "-o", out], check=True, cwd=os.path.dirname(SRC),
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        kernels, problems = check(open(out).read())
    print(f"{kernels} prefill kernels checked, {len(problems)} uses of an LDS destination before its wait")
    for k, ln, line, bad in problems[:20]:
        print(f"  {k[:60]}… line {ln}: {line}   (v{bad})")
    return 1 if problems or not kernels else 0


if __name__ == "__main__":
    sys.exit(main())
