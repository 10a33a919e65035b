#!/usr/bin/env python3
"""Prints register, scratch and code-size figures per kernel from a device assembly file (hipcc --cuda-device-only -S)."""
import re, sys
name = None
for line in open(sys.argv[1]):
    m = re.match(r"\s*\.size\s+(\S+),", line)
    if m: name = m.group(1)
    m = re.match(r"; (codeLenInByte|NumSgprs|NumVgprs|ScratchSize|Occupancy|LDSByteSize)\s*[:=]\s*(\d+)", line)
    if m and name:
        if m.group(1) == "codeLenInByte":
            short = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", name)[:24]
            print("\n%-26s" % short, end="")
        print(" %s=%s" % (m.group(1), m.group(2)), end="")
print()
