#!/usr/bin/env python
"""`xxd -i -n NAME file out.h` for images without xxd (the reference's CMakeLists.txt:489-512 embeds welle-cli's web
page that way): python tools/xxd_i.py NAME infile outfile"""
import sys

name, src, dst = sys.argv[1:4]
data = open(src, "rb").read()
with open(dst, "w") as f:
    f.write("unsigned char %s[] = {\n" % name)
    for i in range(0, len(data), 12):
        f.write("  " + ", ".join("0x%02x" % b for b in data[i:i + 12]) + (",\n" if i + 12 < len(data) else "\n"))
    f.write("};\nunsigned int %s_len = %d;\n" % (name, len(data)))
