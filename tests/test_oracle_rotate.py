"""Pin oracle/pil_rotate.py (the restatement of Pillow's bicubic Image.rotate that --augment_rotation goes through,
/root/reference/train_galaxy.py:41-54, train_particles.py:31-43): against the images the reference itself fed its
encoder (fixtures written by tests/golden/gen_golden.py) and, where Pillow is importable, against Pillow directly.
CPU only."""
import numpy as np
import pytest

import cases as C
from helpers import load_golden
from oracle import pil_rotate as R


@pytest.mark.parametrize("name", ["galaxy_augment", "particles_augment"])
def test_augmented_batch_is_what_the_reference_fed_its_encoder(name):
    case = C.CASES_BY_NAME[name]
    inp = C.build_inputs(case)
    gold = load_golden(name)
    fn = R.augment_galaxy if case["script"] == "galaxy" else R.augment_particles
    got = fn(inp["y"], inp["offset"])
    assert np.array_equal(got, gold["y_rot"])                                  # bit-exact
    assert not np.array_equal(gold["y_rot"], inp["y"])


def test_restatement_equals_pillow_bit_for_bit():
    Image = pytest.importorskip("PIL.Image")
    rs = np.random.RandomState(3)
    for trial in range(60):
        n = int(rs.choice([5, 8, 17, 28, 40]))
        ang = float(rs.uniform(-30, 400)) if trial % 6 else float(rs.choice([0, 90, 180, 270, 360, 45]))
        c = int(rs.choice([1, 3]))
        img = (rs.uniform(size=(n, n, c)) * 255).astype(np.uint8)
        src = img[:, :, 0] if c == 1 else img
        assert np.array_equal(np.array(Image.fromarray(src).rotate(ang, resample=Image.BICUBIC)), R.rotate_u8(src, ang))
        f = rs.normal(size=(n, n)).astype(np.float32)
        assert np.array_equal(np.array(Image.fromarray(f).rotate(ang, resample=Image.BICUBIC)), R.rotate_f32(f, ang))


def test_non_square_images_take_the_general_path_for_quarter_turns():
    Image = pytest.importorskip("PIL.Image")
    rs = np.random.RandomState(4)
    f = rs.normal(size=(6, 9)).astype(np.float32)
    for ang in (90.0, 270.0, 180.0, 33.0):
        assert np.array_equal(np.array(Image.fromarray(f).rotate(ang, resample=Image.BICUBIC)), R.rotate_f32(f, ang))
