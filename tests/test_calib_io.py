"""Result I/O against the reference's own example file: EpipolarRectify/calib.yaml,
written by cv::FileStorage from main.cpp:305-319, kept as tests/golden/reference_calib.yaml."""
import os

import numpy as np
import pytest

from tscm_calib_amd import calib_io, lib, synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "reference_calib.yaml")


def test_reads_the_reference_example_file():
    intr, Twc = calib_io.read_calib_yaml(GOLDEN)
    assert intr.shape == (4, 9) and Twc.shape == (4, 3, 4)
    assert np.array_equal(intr, synth.CALIB_INTR)         # the literals of calib.yaml:3-59
    assert np.array_equal(Twc, synth.CALIB_TWC)
    assert np.array_equal(Twc[0], np.eye(3, 4))


def test_writer_reproduces_the_reference_file_byte_for_byte():
    text = open(GOLDEN, "rb").read().decode()
    intr, Twc = calib_io.parse_calib_yaml(text)
    out = calib_io.format_calib_yaml(intr, Twc[:, :, :3], Twc[:, :, 3])
    assert out == text


def test_write_read_round_trip_is_exact(tmp_path):
    rng = np.random.default_rng(4)
    C = 8
    intr = rng.normal(size=(C, 9)) * 10.0 ** rng.integers(-8, 8, size=(C, 9))
    intr[0, 7:] = 0.0
    intr[1, 0] = 431.0          # integral values are written as "431."
    intr[1, 1] = -7.0
    R = rng.normal(size=(C, 3, 3))
    t = rng.normal(size=(C, 3)) * 1e3
    path = str(tmp_path / "calib.yaml")
    calib_io.write_calib_yaml(path, intr, R, t)
    intr2, Twc2 = calib_io.read_calib_yaml(path)
    assert np.array_equal(intr2, intr)                    # %.16e round-trips a double exactly
    assert np.array_equal(Twc2[:, :, :3], R) and np.array_equal(Twc2[:, :, 3], t)
    text = open(path).read()
    assert text.startswith("%YAML:1.0\n---\ncam0: !!opencv-matrix\n   rows: 1\n   cols: 9\n   dt: d\n   data: [ ")
    assert " 431., -7.," in text
    assert max(len(l) for l in text.splitlines()) <= 72


def test_non_finite_values_use_the_filestorage_spelling():
    intr = np.zeros((1, 9))
    intr[0, :3] = [np.nan, np.inf, -np.inf]
    text = calib_io.format_calib_yaml(intr, np.eye(3)[None], np.zeros((1, 3)))
    assert "data: [ .Nan, .Inf, -.Inf, 0., 0., 0., 0., 0., 0. ]" in text
    back, _ = calib_io.parse_calib_yaml(text)
    assert np.isnan(back[0, 0]) and back[0, 1] == np.inf and back[0, 2] == -np.inf


def test_parse_errors_are_reported():
    with pytest.raises(lib.TscmError) as e:
        calib_io.parse_calib_yaml("%YAML:1.0\n---\ncam0: !!opencv-matrix\n   rows: 1\n   cols: 9\n   dt: d\n   data: [ 1., 2. ]\n")
    assert e.value.code == -1 and "rows x cols" in str(e.value)
    with pytest.raises(lib.TscmError):
        calib_io.parse_calib_yaml("%YAML:1.0\n---\ncam0: !!opencv-matrix\n   rows: 1\n   cols: 9\n   dt: d\n"
                                  "   data: [ 1., 2., 3., 4., 5., 6., 7., 8., 9. ]\n")       # no Twc0
    with pytest.raises(lib.TscmError):
        calib_io.read_calib_yaml("/nonexistent/calib.yaml")
    intr, Twc = calib_io.parse_calib_yaml("%YAML:1.0\n---\n")
    assert intr.shape == (0, 9)


def test_corner_list_round_trip_is_exact(tmp_path):
    p = synth.make_problem(4, 6, 12)
    inp = synth.make_rig_input(p)
    path = str(tmp_path / "corners.txt")
    calib_io.write_corners(path, inp.has, inp.pix_u, inp.pix_v, 9, 6, 45.0)
    d = calib_io.read_corners(path)
    assert (d["board_cols"], d["board_rows"], d["pitch"], d["image_size"]) == (9, 6, 45.0, (1280, 1080))
    assert np.array_equal(d["has"], inp.has)
    assert np.array_equal(d["pix_u"], inp.pix_u) and np.array_equal(d["pix_v"], inp.pix_v)      # %.17g round-trips doubles
    text = open(path).read().splitlines()
    assert text[0] == "TSCM-CORNERS 1" and text[1].startswith("cameras 4 boards 12 cols 9 rows 6 pitch 45 image 1280 1080")
    assert sum(1 for l in text if l.startswith("view ")) == int(inp.has.sum())


def test_corner_list_errors(tmp_path):
    bad = tmp_path / "bad.txt"
    bad.write_text("TSCM-CORNERS 1\ncameras 1 boards 1 cols 2 rows 2 pitch 10 image 640 480\nview 0 0\n1 2\n3 4\n5 6\n")   # 3 of 4 corners
    with pytest.raises(lib.TscmError) as e:
        calib_io.read_corners(str(bad))
    assert "expected 4 corners" in str(e.value)
    bad.write_text("TSCM-CORNERS 1\ncameras 1 boards 1 cols 2 rows 2 pitch 10 image 640 480\nview 0 3\n")
    with pytest.raises(lib.TscmError):
        calib_io.read_corners(str(bad))
    bad.write_text("something else\n")
    with pytest.raises(lib.TscmError):
        calib_io.read_corners(str(bad))
    with pytest.raises(lib.TscmError):
        calib_io.read_corners(str(tmp_path / "missing.txt"))
    empty = tmp_path / "empty.txt"
    calib_io.write_corners(str(empty), np.zeros((2, 3), dtype=np.uint8), np.zeros((2, 3, 54)), np.zeros((2, 3, 54)), 9, 6, 45.0)
    d = calib_io.read_corners(str(empty))
    assert d["has"].shape == (2, 3) and not d["has"].any()


def test_matrix_node_without_data_is_an_error_not_a_crash():
    """Found by the sanitizer fuzzing: a cam node with rows/cols but no data: used to be copied from an empty vector."""
    text = ("%YAML:1.0\n---\ncam0: !!opencv-matrix\n   rows: 1\n   cols: 9\n   dt: d\n"
            "Twc0: !!opencv-matrix\n   rows: 3\n   cols: 4\n   dt: d\n   data: [ 1., 0., 0., 0., 0., 1., 0., 0., 0., 0., 1., 0. ]\n")
    with pytest.raises(Exception) as e:
        calib_io.parse_calib_yaml(text)
    assert "camera 0" in str(e.value)
