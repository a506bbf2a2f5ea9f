"""wafer-hip's array / potential_sub files in the reference's five formats
(wafer_amd/csrc/wafer_files.h; output.rs:85-400, input.rs:60-720), through
`wafer-hip --convert IN OUT`.  The writers are checked with independent
decoders (python's msgpack / json / yaml), the readers with files those
encoders produce, and every format round-trips bit for bit."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.environ.get("WAFER_CLI_BIN") or os.path.join(ROOT, "wafer_amd", "wafer-hip")   # (tests/test_sanitizers.py: the ASan build)
EXT = ["mpk", "csv", "json", "yaml", "ron"]


@pytest.fixture(scope="module")
def cli():
    if not os.path.exists(CLI):
        from wafer_amd import build
        build.build()
    return CLI


def convert(cli, src, dst):
    r = subprocess.run([cli, "--convert", str(src), str(dst)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return dst


def sample(shape=(3, 4, 5), seed=5):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(shape)
    a.flat[0] = 1.0            # integral value: "1.0" in json / csv / yaml
    a.flat[1] = 1.25e-7        # exponent notation below 1e-5
    a.flat[2] = -3.5e21        # and above 1e16 / 1e21
    a.flat[3] = 0.0
    a.flat[4] = 5e-324         # smallest denormal
    a.flat[5] = 123456.789
    return a


def write_csv(path, a):
    with open(path, "w") as f:
        for (i, j, k), v in np.ndenumerate(a):
            f.write(f"{i},{j},{k},{float(v)!r}\n")


def test_messagepack_writer_layout(cli, tmp_path):
    """rmp-serde 0.13 compact form of ndarray's {v, dim, data}: [1, [nx, ny, nz], [f64...]]"""
    import msgpack
    a = sample()
    write_csv(tmp_path / "a.csv", a)
    raw = open(convert(cli, tmp_path / "a.csv", tmp_path / "a.mpk"), "rb").read()
    v, dim, data = msgpack.unpackb(raw)
    assert v == 1 and dim == [3, 4, 5]
    assert np.array_equal(np.array(data).reshape(3, 4, 5), a)
    # fixarray(3), fixint 1, fixarray(3) + three fixints, array16 of 60, then 0xcb + 8 big-endian bytes each
    assert raw[:6] == bytes([0x93, 0x01, 0x93, 3, 4, 5]) and raw[6:9] == bytes([0xdc, 0, 60])
    assert raw[9] == 0xcb and struct.unpack(">d", raw[10:18])[0] == a.flat[0]
    assert len(raw) == 9 + 60 * 9


def test_messagepack_reader_accepts_array_and_map_forms(cli, tmp_path):
    import msgpack
    a = sample((2, 3, 4), seed=8)
    forms = {
        "compact": [1, [2, 3, 4], a.ravel().tolist()],
        "named": {"v": 1, "dim": [2, 3, 4], "data": a.ravel().tolist()},
    }
    for name, obj in forms.items():
        (tmp_path / f"{name}.mpk").write_bytes(msgpack.packb(obj, use_single_float=False))
        back = np.loadtxt(convert(cli, tmp_path / f"{name}.mpk", tmp_path / f"{name}.csv"), delimiter=",")
        assert np.array_equal(back[:, 3].reshape(2, 3, 4), a)
        assert np.array_equal(back[:3, :3], [[0, 0, 0], [0, 0, 1], [0, 0, 2]])
    # integers and float32 payloads (other writers) are widened
    (tmp_path / "ints.mpk").write_bytes(msgpack.packb([1, [1, 1, 3], [1, -2, 70000]]))
    back = np.loadtxt(convert(cli, tmp_path / "ints.mpk", tmp_path / "ints.csv"), delimiter=",")
    assert back[:, 3].tolist() == [1.0, -2.0, 70000.0]


def test_json_and_yaml_writers_decode_to_the_same_array(cli, tmp_path):
    import yaml
    a = sample()
    write_csv(tmp_path / "a.csv", a)
    j = json.load(open(convert(cli, tmp_path / "a.csv", tmp_path / "a.json")))
    assert j["v"] == 1 and j["dim"] == [3, 4, 5] and np.array_equal(np.array(j["data"]).reshape(3, 4, 5), a)
    text = open(tmp_path / "a.json").read()
    assert text.startswith('{\n  "v": 1,\n  "dim": [\n    3,\n    4,\n    5\n  ],\n  "data": [\n    1.0,\n    1.25e-7,\n    -3.5e21,\n    0.0,\n    5e-324,\n    123456.789,')
    assert text.endswith("\n  ]\n}")           # serde_json::to_writer_pretty
    y = yaml.safe_load(open(convert(cli, tmp_path / "a.csv", tmp_path / "a.yaml")))
    assert y["v"] == 1 and y["dim"] == [3, 4, 5]
    assert np.array_equal(np.array(y["data"], dtype=float).reshape(3, 4, 5), a)
    assert open(tmp_path / "a.yaml").read().startswith("---\nv: 1\ndim:\n  - 3\n  - 4\n  - 5\ndata:\n  - 1.0\n  - 1.25e-7\n")


@pytest.mark.parametrize("src", EXT)
@pytest.mark.parametrize("dst", EXT)
def test_round_trip_every_pair_of_formats(cli, tmp_path, src, dst):
    a = sample((4, 2, 3), seed=11)
    write_csv(tmp_path / "seed.csv", a)
    first = convert(cli, tmp_path / "seed.csv", tmp_path / f"first.{src}")
    second = convert(cli, first, tmp_path / f"second.{dst}")
    back = np.loadtxt(convert(cli, second, tmp_path / "back.csv"), delimiter=",")
    assert np.array_equal(back[:, 3].reshape(4, 2, 3), a)


def test_readers_accept_foreign_layouts(cli, tmp_path):
    """compact json, flow-style yaml, ron without pretty printing"""
    a = np.arange(1, 9, dtype=float).reshape(2, 2, 2) / 8
    body = ",".join(repr(float(v)) for v in a.ravel())
    (tmp_path / "c.json").write_text('{"v":1,"dim":[2,2,2],"data":[%s]}' % body)
    (tmp_path / "c.yaml").write_text("v: 1\ndim: [2, 2, 2]\ndata: [%s]\n" % body)
    (tmp_path / "c.ron").write_text("(v:1,dim:(2,2,2),data:[%s])" % body)
    for ext in ("json", "yaml", "ron"):
        back = np.loadtxt(convert(cli, tmp_path / f"c.{ext}", tmp_path / f"{ext}.csv"), delimiter=",")
        assert np.array_equal(back[:, 3].reshape(2, 2, 2), a)


def test_singular_potential_sub_files(cli, tmp_path):
    """PotentialSubSingle {pot_sub} (output.rs:224-377, input.rs:304-470)"""
    import msgpack
    import yaml
    (tmp_path / "s.csv").write_text("213.5\n")
    assert msgpack.unpackb(open(convert(cli, tmp_path / "s.csv", tmp_path / "s.mpk"), "rb").read()) == [213.5]
    assert json.load(open(convert(cli, tmp_path / "s.mpk", tmp_path / "s.json"))) == {"pot_sub": 213.5}
    assert yaml.safe_load(open(convert(cli, tmp_path / "s.json", tmp_path / "s.yaml"))) == {"pot_sub": 213.5}
    convert(cli, tmp_path / "s.yaml", tmp_path / "s.ron")
    assert "pot_sub: 213.5" in open(tmp_path / "s.ron").read()
    assert open(convert(cli, tmp_path / "s.ron", tmp_path / "back.csv")).read().strip() == "213.5"
    (tmp_path / "named.mpk").write_bytes(msgpack.packb({"pot_sub": 7.25}))
    assert open(convert(cli, tmp_path / "named.mpk", tmp_path / "named.csv")).read().strip() == "7.25"


def test_malformed_files_fail_loudly(cli, tmp_path):
    (tmp_path / "short.json").write_text('{"v":1,"dim":[2,2,2],"data":[1.0,2.0]}')
    r = subprocess.run([cli, "--convert", str(tmp_path / "short.json"), str(tmp_path / "x.csv")], capture_output=True, text=True)
    assert r.returncode == 1 and "ArrayShape" in r.stderr
    (tmp_path / "bad.mpk").write_bytes(b"\x93\x02\x93\x01\x01\x01\x91\xcb" + b"\0" * 8)   # format version 2
    r = subprocess.run([cli, "--convert", str(tmp_path / "bad.mpk"), str(tmp_path / "x.csv")], capture_output=True, text=True)
    assert r.returncode == 1 and "version" in r.stderr
    (tmp_path / "rec.csv").write_text("0,0,0,1.0\n0,0,x,2.0\n")
    r = subprocess.run([cli, "--convert", str(tmp_path / "rec.csv"), str(tmp_path / "x.json")], capture_output=True, text=True)
    assert r.returncode == 1 and "ParsePlainRecord" in r.stderr
    r = subprocess.run([cli, "--convert", str(tmp_path / "missing.mpk"), str(tmp_path / "x.json")], capture_output=True, text=True)
    assert r.returncode == 1 and "FileNotFound" in r.stderr


def test_reference_output_potential_sub_cases(cli, tmp_path):
    """the inputs of the reference's own `output_potential_sub` test (output.rs:799-821: it only
    asserts is_ok): singular values 213.0 / 21.0 / 24.8 / 29.1 / 94.32 and a zeros((2, 2, 2)) array,
    in every format -- written, read back, identical"""
    import msgpack
    for ext, val in zip(EXT, [213.0, 21.0, 24.8, 29.1, 94.32]):
        (tmp_path / "v.json").write_text('{"pot_sub": %r}' % val)
        out = convert(cli, tmp_path / "v.json", tmp_path / f"test.{ext}")
        back = json.load(open(convert(cli, out, tmp_path / f"back_{ext}.json")))
        assert back == {"pot_sub": val}
    assert msgpack.unpackb(open(tmp_path / "test.mpk", "rb").read()) == [213.0]
    assert open(tmp_path / "test.csv").read() == "21\n"                 # R64::to_string(): Display, no ".0"
    write_csv(tmp_path / "z.csv", np.zeros((2, 2, 2)))
    for ext in EXT:
        out = convert(cli, tmp_path / "z.csv", tmp_path / f"zeros.{ext}")
        back = np.loadtxt(convert(cli, out, tmp_path / f"zeros_back_{ext}.csv"), delimiter=",")
        assert back.shape == (8, 4) and not back[:, 3].any()


def test_multi_rank_driver_stages_input_arrays(tmp_path):
    """wafer_amd.run.staged_array: rank 0 converts ./input/<stem>.* once into a framed .npy (wafer-hip
    --convert ... --pad E, no GPU involved), every rank memory-maps it; the cache follows the source's
    modification time; the configured file type arbitrates between several files (input.rs:75-110)"""
    import json
    import time
    from wafer_amd import run
    rng = np.random.default_rng(4)
    a = rng.standard_normal((5, 4, 6))
    inp = tmp_path / "input"
    inp.mkdir()
    (inp / "potential.json").write_text(json.dumps({"v": 1, "dim": list(a.shape), "data": a.ravel().tolist()}))
    assert run.staged_array(str(inp), "wavefunction_0", "Csv", 1, 0) is None
    m = run.staged_array(str(inp), "potential", "Csv", 2, 0)
    assert m.shape == (9, 8, 10) and np.array_equal(m[2:-2, 2:-2, 2:-2], a) and np.abs(m).sum() == np.abs(a).sum()
    cache = inp / ".wafer_amd" / "potential.pad2.npy"
    stamp = cache.stat().st_mtime_ns
    again = run.staged_array(str(inp), "potential", "Csv", 2, 1, wait_s=5)     # another rank: waits, never converts
    assert np.array_equal(again, m) and cache.stat().st_mtime_ns == stamp
    # a second file: the configured type decides, and a newer source invalidates the cache
    b = a + 1.0
    time.sleep(0.05)
    with open(inp / "potential.csv", "w") as f:
        for i in range(5):
            for j in range(4):
                for k in range(6):
                    f.write(f"{i},{j},{k},{float(b[i, j, k])!r}\n")
    assert run.find_input(str(inp), "potential", "Csv").endswith("potential.csv")
    assert run.find_input(str(inp), "potential", "Json").endswith("potential.json")
    assert run.find_input(str(inp), "potential", "Yaml").endswith("potential.csv")     # neither is yaml: the first in mpk, csv, json, ... order
    m2 = run.staged_array(str(inp), "potential", "Csv", 2, 0)
    assert np.array_equal(m2[2:-2, 2:-2, 2:-2], b)
    # a single value (potential_sub) becomes a 0-d array
    (inp / "potential_sub.yaml").write_text("---\npot_sub: 2.5\n")
    s = run.staged_array(str(inp), "potential_sub", "Csv", 0, 0)
    assert s.ndim == 0 and float(s) == 2.5
    # a ready-made .npy is used as it is
    np.save(inp / "wavefunction_3.npy", np.ones((3, 3, 3)))
    assert run.staged_array(str(inp), "wavefunction_3", "Csv", 1, 0).shape == (3, 3, 3)
