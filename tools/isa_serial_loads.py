"""Lists kernels whose ISA has many `load -> s_waitcnt vmcnt(0)` pairs a few instructions apart: the signature of an
epilogue (or any unrolled loop) in which every load waits alone for memory because the compiler may not move it above the
previous iteration's store.  Usage: python tools/isa_serial_loads.py [file.hip ...]  (default: every csrc/*.hip)."""
import glob, os, re, subprocess, sys, tempfile

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "unlearn_saliency_amd", "csrc")
FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fvisibility=hidden -S --cuda-device-only".split()


def kernels(asm):
    name, body = None, []
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
        elif name is not None:
            if line.startswith(".Lfunc_end"):
                yield name, body
                name = None
            else:
                t = line.strip()
                if t and not t.startswith((";", ".")):
                    body.append(t)


def score(body):
    n = 0
    for i, t in enumerate(body):
        if re.match(r"(global|buffer|flat)_load", t):
            for u in body[i + 1:i + 4]:
                if re.match(r"(global|buffer|flat)_(load|store)", u):
                    break
                if u.startswith("s_waitcnt") and "vmcnt(0)" in u:
                    n += 1
                    break
    return n


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    rows = []
    for f in map(os.path.abspath, files):
        with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
            subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, f, "-o", tmp.name], check=True, stderr=subprocess.DEVNULL, cwd=os.path.dirname(f))
            asm = open(tmp.name).read()
        for name, body in kernels(asm):
            s = score(body)
            if s >= 6:
                d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                rows.append((s, os.path.basename(f), d.replace("(anonymous namespace)::", "")[:110]))
    for s, f, d in sorted(rows, reverse=True):
        print(f"{s:5d}  {f:22s} {d}")
    if not rows:
        print("no kernel with 6 or more lone load -> vmcnt(0) pairs")


if __name__ == "__main__":
    main()
