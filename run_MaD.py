#!/usr/bin/env python3
"""Example driver, same calls as the reference's run_MaD.py:64-76.

    python run_MaD.py <map.mrc|map.sit|map.pdb> <resolution> <subunit.pdb>[:n_copies] [more subunits...]

Without arguments it docks a small synthetic dimer (written to ./synthetic_example) so
that the flow can be tried without the lab's data set.  Needs an MI355X: the hot path
has no CPU fallback.
"""
import os
import sys

from mad import MaD


def _synthetic_example(folder="synthetic_example"):
    import numpy as np
    from mad_amd import synth
    os.makedirs(folder, exist_ok=True)
    rng = np.random.default_rng(7)
    coords, names, elems = synth.random_globule(1500, 16.0, seed=1)
    sub = os.path.join(folder, "subunit.pdb")
    synth.write_pdb(sub, coords, names, elems)
    parts = [synth.place(coords, synth.random_rotation(rng), t) for t in ([0, 0, 0], [38, 6, -4])]
    asm = os.path.join(folder, "assembly.pdb")
    synth.write_pdb(asm, np.concatenate(parts), names * 2, elems * 2)
    return asm, 10.0, [(sub, 2)]


if __name__ == "__main__":
    if len(sys.argv) >= 4:
        map_file, resolution = sys.argv[1], float(sys.argv[2])
        subunits = [(a.split(":")[0], int(a.split(":")[1]) if ":" in a else 1) for a in sys.argv[3:]]
    else:
        map_file, resolution, subunits = _synthetic_example()

    # Make a MaD instance
    mad = MaD.MaD()

    # Add map, specify its resolution
    mad.add_map(map_file, resolution)

    # Add components
    for path, n_copies in subunits:
        mad.add_subunit(path, n_copies=n_copies)

    # Get solutions per component
    mad.run()

    # Build assembly models from solutions
    mad.build_assembly()
