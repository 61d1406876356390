"""Compile every HIP source of the library for gfx950 (device code only, to assembly) and report, per kernel, the code
size and how many global loads are immediately followed by `s_waitcnt vmcnt(0)` -- the signature of loads that hipcc
issued one at a time (typically `v = cond ? p[i] : 0` per element of an unrolled batch).
Runs without a GPU.   python tools/isa_scan.py [min_serialised=4]"""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vmlmf_amd", "csrc")


def scan(path, tmp):
    out = os.path.join(tmp, os.path.basename(path) + ".s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
                    "--cuda-device-only", "-S", path, "-o", out], check=True, stderr=subprocess.DEVNULL)
    txt = open(out).read()
    size = dict(re.findall(r"^(_Z\S+):.*?; codeLenInByte = (\d+)", txt, flags=re.M | re.S))
    parts = re.split(r"^(_Z\S+):[^\n]*\n", txt, flags=re.M)
    rows = []
    for i in range(1, len(parts), 2):
        name, body = parts[i], parts[i + 1].split("s_endpgm")[0]
        seq = []
        for line in body.split("\n"):
            t = line.split()
            if not t:
                continue
            if t[0].startswith("global_load") and "lds" not in line:
                seq.append("L")
            elif t[0] == "s_waitcnt" and "vmcnt(0)" in line:
                seq.append("W0")
        serial = sum(1 for a, b in zip(seq, seq[1:]) if a == "L" and b == "W0")
        m = re.search(r"; codeLenInByte = (\d+)", parts[i + 1])
        rows.append((serial, seq.count("L"), int(m.group(1)) if m else 0, name))
    return rows


def main():
    thresh = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    with tempfile.TemporaryDirectory() as tmp:
        for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
            rows = [r for r in scan(src, tmp) if r[0] >= thresh or r[2] > 60000]
            if rows:
                print(os.path.basename(src))
                for serial, loads, code, name in sorted(rows, reverse=True):
                    demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
                    print(f"   {serial:3d} of {loads:3d} loads wait vmcnt(0)   {code:6d} B   {demangled[:90]}")


if __name__ == "__main__":
    main()
