"""Extract the gfx950 code objects of a libsspgpu.so (clang offload bundles in .hip_fatbin) and disassemble the kernels whose name
contains a substring:   python tools/extract_co.py lib.so kernel-substring out-dir   -> out-dir/<n>.co, out-dir/<kernel>.s"""
import os, struct, subprocess, sys
lib, sub, out = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(out, exist_ok=True)
data = open(lib, "rb").read()
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
pos, n = 0, 0
while True:
    i = data.find(MAGIC, pos)
    if i < 0:
        break
    nb = struct.unpack_from("<Q", data, i + 24)[0]
    p = i + 32
    for _ in range(nb):
        off, size, tl = struct.unpack_from("<QQQ", data, p)
        triple = data[p + 24:p + 24 + tl].decode()
        p += 24 + tl
        if "gfx950" in triple and size:
            co = os.path.join(out, "%d.co" % n)
            open(co, "wb").write(data[i + off:i + off + size])
            syms = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-sW", co], capture_output=True, text=True).stdout
            for line in syms.splitlines():
                f = line.split()
                parts = [f[1], "T", f[7]] if len(f) == 8 and f[3] == "FUNC" else []
                if len(parts) == 3 and parts[1] in "Tt" and sub in parts[2]:
                    d = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", "--symbolize-operands", "--disassemble-symbols=" + parts[2], co],
                                       capture_output=True, text=True).stdout
                    open(os.path.join(out, parts[2][:120] + ".s"), "w").write(d)
                    print(co, parts[2][:100], len(d.splitlines()))
            n += 1
    pos = i + 24
