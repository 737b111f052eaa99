#!/usr/bin/env python3
"""Extract the 256 rBRIEF point pairs (1024 ints) used by the reference extractor.

The table is constant data (`bit_pattern_31_`, /root/reference/src/ORBextractor.cc:150-408).
Only the numbers are taken; they are re-emitted 16 per line (= 4 point pairs) into
include/orb_pattern_31.inc, which both the HIP library and the CPU oracle include.

Usage: python tools/gen_pattern.py [/root/reference/src/ORBextractor.cc]
"""
import re
import sys
import zlib
from pathlib import Path

src = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/ORBextractor.cc")
text = src.read_text(errors="replace")
start = text.index("bit_pattern_31_[256*4]")
body = text[text.index("{", start) + 1: text.index("};", start)]
body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
vals = [int(v) for v in re.findall(r"-?\d+", body)]
assert len(vals) == 1024, len(vals)
assert all(-13 <= v <= 13 for v in vals)
out = Path(__file__).resolve().parent.parent / "include" / "orb_pattern_31.inc"
lines = ["/* 256 rBRIEF test pairs (x0,y0,x1,y1), 4 pairs per line; all coordinates in [-13,13].",
         " * Constant data of the reference extractor (src/ORBextractor.cc:150-408).",
         " * crc32 of the 1024 values as int8 bytes: 0x%08x */" % zlib.crc32(bytes((v & 0xFF) for v in vals))]
for i in range(0, 1024, 16):
    lines.append(",".join("%d" % v for v in vals[i:i + 16]) + ("," if i + 16 < 1024 else ""))
out.write_text("\n".join(lines) + "\n")
print("wrote", out, "crc32=0x%08x" % zlib.crc32(bytes((v & 0xFF) for v in vals)))
