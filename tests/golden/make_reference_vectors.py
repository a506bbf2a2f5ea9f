#!/usr/bin/env python3
"""Extracts the known-answer DATA of the reference's own unit tests into
reference_unit_vectors.json.  Run in the build container only (it reads
/root/reference, which does not exist on the GPU box); the JSON is committed.

Only numbers are taken: inputs are described by their generating rule, and
expected outputs are the literals the reference's asserts compare against.
"""
import json
import os
import re

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))


def r64_literals(path, lo, hi):
    """all r64(<number>) literals between 1-based lines lo..hi"""
    with open(path) as f:
        lines = f.readlines()[lo - 1:hi]
    return [float(m) for m in re.findall(r"r64\((-?[0-9.]+(?:e-?[0-9]+)?)\)", "".join(lines))]


def main():
    out = {}
    # grid.rs:721-746 gram_schmidt
    gs = r64_literals(f"{REF}/grid.rs", 732, 744)
    assert len(gs) == 8
    out["gram_schmidt"] = {
        "cite": "grid.rs:721-746",
        "shape": [2, 2, 2],
        "lower_rule": "i+j+k",
        "phi_rule": "-(i+j+k)",
        "expected": gs,
        "tol": 0.01,
    }
    # grid.rs:748-778 work area
    out["work_area"] = {"cite": "grid.rs:748-778", "shape": [5, 8, 7], "ext": 1,
                        "expected_dims": [3, 6, 5]}
    # grid.rs:780-786 norm2
    out["norm2"] = {"cite": "grid.rs:780-786", "shape": [5, 8, 7], "ext": 1,
                    "phi_rule": "i*j*k", "expected": r64_literals(f"{REF}/grid.rs", 785, 785)[0],
                    "eps": 1.0e-6}
    # grid.rs:788-799 wfn_normalise
    lits = r64_literals(f"{REF}/grid.rs", 788, 799)
    out["wfn_normalise"] = {"cite": "grid.rs:788-799", "shape": [3, 2, 5], "phi_rule": "i*j*k",
                            "norm2": lits[1], "expected_divisor": lits[0], "tol": 0.01}
    assert out["wfn_normalise"]["norm2"] == 1.23 and out["wfn_normalise"]["expected_divisor"] == 1.1091
    # potential.rs:434-454
    out["distance_squared"] = {"cite": "potential.rs:434-443", "size": [5, 6, 3], "idx": [3, 3, 3],
                               "expected": 1.25, "eps": 1.0e-6}
    with open(f"{REF}/potential.rs") as f:
        src = f.read()
    m = re.search(r"alphas\(md\), ([0-9.]+), (1e-14)", src)
    out["running_coupling"] = {"cite": "potential.rs:445-449", "mu": 3.2,
                               "expected": float(m.group(1)), "eps": float(m.group(2))}
    m = re.search(r"mu\(t\), ([0-9.]+), (1e-14)", src)
    out["debye_screening_mass"] = {"cite": "potential.rs:450-454", "t": 5.2,
                                   "expected": float(m.group(1)), "eps": float(m.group(2))}
    # input.rs:732-824 interpolation (assert_eq!: exact)
    src_vals = r64_literals(f"{REF}/input.rs", 733, 747)
    exp = r64_literals(f"{REF}/input.rs", 756, 821)
    assert len(src_vals) == 8 and len(exp) == 64
    out["interpolation"] = {"cite": "input.rs:732-824", "source_shape": [2, 2, 2],
                            "source": src_vals, "target_shape": [4, 4, 4], "expected": exp,
                            "exact": True}
    # output.rs:758-762 directory_string
    with open(f"{REF}/output.rs") as f:
        osrc = f.read()
    m = re.search(r'let bad_string = "(.*)";\s*assert_eq!\(sanitize_string\(&bad_string\), "(.*)"\);', osrc)
    out["directory_string"] = {"cite": "output.rs:758-762",
                               "input": m.group(1).encode().decode("unicode_escape"), "expected": m.group(2)}
    with open(os.path.join(HERE, "reference_unit_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out), "vectors")


if __name__ == "__main__":
    main()
