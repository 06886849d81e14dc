"""Measurement legs of bench.py (repo root).  bench.py itself holds the argument parsing, the headline's timed region,
the CPU baseline / parity leg (the only place outside tests/ that runs oracle/) and the JSON line; everything else is
here:

    common.py    workload constants (BASELINE configs[1]), model / input construction shared with the tests
    k1.py        roofline probes of the cost-volume build: HIP-event timing of the C-ABI launches, the same-run fill /
    copy
                 ceilings, the beyond-the-Infinity-Cache launch, the contracted first layer (`roofline.fused`)
    extras.py    one pass at a time, the f32-only engine, concurrent lanes, temporal sequences
    training.py  the data-parallel training step (`--mode train | train-graph`, `training` of the default line)
"""
