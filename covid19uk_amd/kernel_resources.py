"""The compiler's account of the library's kernels: parser of `hipcc -Rpass-analysis=kernel-resource-usage` remarks
(registers, scratch, spills, LDS, occupancy per kernel).  `__graft_entry__.build()` keeps the result next to the library as
kernel_resources.json; tests/test_resources.py holds the hot-path instances to it; tools/dev/resources.py prints it."""
import re


def demangle(name):
    """`_ZN4seir6k_leapILi1ELi6ELi1ELi6EEEv...` -> `k_leap<1,6,1,6>` (the library's kernels only take integer and bool
    template arguments; anything else is returned as it is)."""
    m = re.match(r"_ZN4seir(\d+)", name)
    if not m:
        m = re.match(r"_Z(\d+)", name)
        if not m:
            return name
    n = int(m.group(1))
    base = name[m.end():m.end() + n]
    rest = name[m.end() + n:]
    if not rest.startswith("I"):
        return base
    args, i = [], 1
    while i < len(rest) and rest[i] != "E":
        a = re.match(r"L([ibjlmxy])(n?)(\d+)E", rest[i:])
        if not a:
            return base + "<?>"
        v = ("-" if a.group(2) else "") + a.group(3)
        args.append({"b": {"0": "false", "1": "true"}.get(v, v)}.get(a.group(1), v) if a.group(1) == "b" else v)
        i += a.end()
    return f"{base}<{','.join(args)}>"


FIELDS = {"sgpr": "TotalSGPRs", "vgpr": "VGPRs", "agpr": "AGPRs", "scratch_bytes_per_lane": r"ScratchSize \[bytes/lane\]",
          "occupancy_waves_per_simd": r"Occupancy \[waves/SIMD\]", "sgpr_spill": "SGPRs Spill", "vgpr_spill": "VGPRs Spill",
          "lds_bytes_per_block": r"LDS Size \[bytes/block\]"}


def parse(text):
    out = {}
    for blk in re.split(r"remark: [^\n]*Function Name: ", text)[1:]:
        name = demangle(blk.split()[0])
        ent = {}
        for key, pat in FIELDS.items():
            m = re.search(r"remark:\s+" + pat + r": (\d+)", blk)
            ent[key] = int(m.group(1)) if m else None
        out[name] = ent
    return out
