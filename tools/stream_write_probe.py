#!/usr/bin/env python3
"""Which decomposition of a many-stream output reaches HBM write bandwidth on MI355X? (round 4)

The witness generator writes 23 GB (slice 1 at 4096 `step` proofs) as 569 000 concurrent output streams of 5 055 words, each advancing 16 bytes
between ~50 multiply-adds of arithmetic: 1.2 TB/s. This probe writes the same volume with the same amount of dependent arithmetic per byte and
varies ONLY the pattern: lanes per stream, contiguous words per lane and step, streams in flight.   python tools/stream_write_probe.py
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent / "probe"))
import gpv_probe as P  # noqa: E402

TOTAL_WORDS = 4096 * 702670          # slice 1 of 4096 `step` proofs
BASE_STREAMS, SPIN = 4096 * 139, 46  # one lane per (proof, permutation); SPIN multiply-adds per 16 bytes ~ the compute-only time of the real kernel


def run(label, n_streams, lps, cw, SPIN=SPIN):
    words_per_stream = TOTAL_WORDS // n_streams
    steps = words_per_stream // (lps * cw)
    stride = (steps * lps * cw + 37 * 2 + 1) // 2 * 2   # streams back to back, a ragged gap between them (like the trace's slices)
    spin = SPIN * cw // 2
    t0 = P.stream_write(0, n_streams, stride, steps, lps, cw, spin)
    t1 = P.stream_write(1, n_streams, stride, steps, lps, cw, spin)
    gb = n_streams * steps * lps * cw * 8 / 1e9
    print("%-78s %8d streams  %5.1f GB  compute only %6.2f ms  with stores %6.2f ms  %5.2f TB/s" % (label, n_streams, gb, t0, t1, gb / t1), flush=True)


print("# tools/stream_write_probe.py (MI355X): %d words, %d multiply-adds of dependent arithmetic per 16 bytes in every row" % (TOTAL_WORDS, SPIN))
run("one lane per stream, 16 B per step (the shipped witness kernels' pattern)", BASE_STREAMS, 1, 2)
run("one lane per stream, 64 B per step (4 stores back to back)", BASE_STREAMS, 1, 8)
run("one lane per stream, 128 B per step (a whole line, 8 stores back to back)", BASE_STREAMS, 1, 16)
run("one lane per stream, 512 B per step", BASE_STREAMS, 1, 64)
run("8 lanes per stream, 16 B each (128 B contiguous per store instruction)", BASE_STREAMS // 8, 8, 2)
run("8 lanes per stream, 128 B each (1 KB contiguous per step)", BASE_STREAMS // 8, 8, 16)
run("64 lanes per stream, 16 B each (1 KB contiguous per store instruction)", BASE_STREAMS // 64, 64, 2)
run("64 lanes per stream, 128 B each (8 KB contiguous per step)", BASE_STREAMS // 64, 64, 16)
run("one lane per stream, 16 B per step, 1/8 of the streams (8 x longer)", BASE_STREAMS // 8, 1, 2)
run("one lane per stream, 16 B per step, 1/64 of the streams", BASE_STREAMS // 64, 1, 2)
# the ceiling: the same volume with NO arithmetic between the stores (what the memory system takes when nothing else is in the way)
run("no arithmetic: 64 lanes per stream, 16 B each (a plain streaming fill)", BASE_STREAMS // 64, 64, 2, SPIN=0)
run("no arithmetic: 8 lanes per stream, 16 B each (the flush events' pattern)", BASE_STREAMS // 8, 8, 2, SPIN=0)
run("no arithmetic: one lane per stream, 16 B per step", BASE_STREAMS, 1, 2, SPIN=0)
