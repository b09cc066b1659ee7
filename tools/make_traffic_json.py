#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of tools/profile_round.sh -- written ON THE GPU BOX by that script, for the library that was
profiled: the file carries libgpv.so's GNU build id, and bench.py quotes it only when the library it loaded carries the same id
(the round-3 line quoted traffic measured before the last kernel change).

    python tools/make_traffic_json.py <dir with pmc_fetch.txt pmc_write.txt> <tag> <fixture> <proofs_per_gpu> > traffic.json

HBM bytes per launch = FETCH_SIZE[KB] x 1024 x 2 (gfx950 half-count correction of /opt/skills/guides/MI355X_MICROARCH.md, calibrated on
k_range_check: 629.2 MB streamed -> 322 132.6 KB reported) + WRITE_SIZE[KB] x 1024.
"""
import json
import struct
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def build_id(path):
    """GNU build id (.note.gnu.build-id) of an ELF64 little-endian shared object, hex; None if it has none."""
    data = Path(path).read_bytes()
    if data[:4] != b"\x7fELF" or data[4] != 2 or data[5] != 1:
        return None
    e_phoff, = struct.unpack_from("<Q", data, 0x20)
    e_phentsize, e_phnum = struct.unpack_from("<HH", data, 0x36)
    for i in range(e_phnum):
        off = e_phoff + i * e_phentsize
        p_type, = struct.unpack_from("<I", data, off)
        if p_type != 4:  # PT_NOTE
            continue
        p_offset, = struct.unpack_from("<Q", data, off + 8)
        p_filesz, = struct.unpack_from("<Q", data, off + 32)
        pos, end = p_offset, p_offset + p_filesz
        while pos + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", data, pos)
            name = data[pos + 12:pos + 12 + namesz]
            desc_off = pos + 12 + (namesz + 3) // 4 * 4
            if ntype == 3 and name.rstrip(b"\0") == b"GNU":
                return data[desc_off:desc_off + descsz].hex()
            pos = desc_off + (descsz + 3) // 4 * 4
    return None


def per_dispatch(path, counter):
    out = {}
    for line in Path(path).read_text().splitlines():
        f = line.split()
        if len(f) >= 6 and f[1] == counter:
            out[f[0]] = (int(f[2]), float(f[4]), float(f[5]))
    return out


def counter_sums(path, counter):
    """{kernel: sum over all rows} of one counter in a pmc summary (SQ counters come as one row per shader engine and dispatch)"""
    out = {}
    try:
        for line in Path(path).read_text().splitlines():
            f = line.split()
            if len(f) >= 6 and f[1] == counter:
                out[f[0]] = float(f[3])
    except OSError:
        pass
    return out


def main():
    d, tag, fixture, n = Path(sys.argv[1]), sys.argv[2], sys.argv[3], int(sys.argv[4])
    fetch, write = per_dispatch(d / "pmc_fetch.txt", "FETCH_SIZE"), per_dispatch(d / "pmc_write.txt", "WRITE_SIZE")
    valu, gui = counter_sums(d / "pmc_sq.txt", "SQ_INSTS_VALU"), counter_sums(d / "pmc_sq.txt", "GRBM_GUI_ACTIVE")
    out = {"_comment": "HBM bytes per launch from separate rocprofv3 --pmc passes (tools/profile_round.sh): FETCH_SIZE[KB] x 1024 x 2 (gfx950 "
                       "half-count correction) + WRITE_SIZE[KB] x 1024. bench.py quotes an entry only for the matching fixture / batch AND the "
                       "library build it was measured on.",
           "_build_id": build_id(ROOT / "gnark-plonky2-verifier_amd" / "libgpv.so"), "_tag": tag}
    for k in sorted(set(fetch) & set(write)):
        if not k.startswith("k_"):
            continue
        out[k] = {"fixture": fixture, "proofs_per_gpu": n, "dispatches_per_step": fetch[k][0] // 2, "fetch_size_kb_raw": fetch[k][1],
                  "write_size_kb": write[k][1], "traffic_bytes_per_launch": int(fetch[k][1] * 1024 * 2 + write[k][1] * 1024),
                  "source": "profiles/%s_pmc_fetch.txt + profiles/%s_pmc_write.txt" % (tag, tag)}
        if k in valu:  # VALU wave-instructions per launch: the SQ pass ran the same number of launches as the FETCH pass (fetch[k][0])
            out[k]["valu_wave_insts_per_launch"] = valu[k] / fetch[k][0]
            out[k]["valu_source"] = "profiles/%s_pmc_sq.txt: sum of SQ_INSTS_VALU / %d launches" % (tag, fetch[k][0])
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
