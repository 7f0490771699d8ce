"""Batch-assembly oracle (oracle/data_oracle.py) against the golden vectors generated from the reference's
StackInputsProcessor / CutMix (tests/golden/data_pipeline.npz, oracle/make_golden_data.py), and the host-side draw logic of
sensorium_amd/data_gpu.py against the oracle.  CPU only."""
import numpy as np
import pytest

from oracle import data_oracle as dorc


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(golden_dir / "data_pipeline.npz")


def _case(gold, c):
    h0, w0, sw, sh, e0, e1, size, step = (int(v) for v in gold[f"c{c}_meta"])
    trials = [dict(video=gold[f"c{c}_video{i}"], behavior=gold[f"c{c}_beh{i}"], pupil_center=gold[f"c{c}_pup{i}"],
                   responses=gold[f"c{c}_resp{i}"]) for i in range(2)]
    return dict(h0=h0, w0=w0, size=(sw, sh), ends=(e0, e1), window=(size, step), fill=float(gold[f"c{c}_fill"]),
                trials=trials)


def test_oracle_matches_reference_stack_inputs_and_targets(gold):
    for c in range(int(gold["num_cases"])):
        k = _case(gold, c)
        d = k["trials"][0]
        idx = dorc.window_indexes(k["ends"][0], *k["window"])
        x = dorc.stack_inputs(d["video"][..., idx], d["behavior"][..., idx], d["pupil_center"][..., idx], k["size"],
                              k["fill"])
        assert x.dtype == np.float32 and np.array_equal(x, gold[f"c{c}_x0"])
        assert np.array_equal(dorc.responses_to_target(d["responses"][..., idx]), gold[f"c{c}_t0"])


def test_host_inputs_processor_matches_reference(gold):
    """sensorium_amd.inputs.StackInputsProcessor (what Predictor builds from a checkpoint's params) against the reference's
    StackInputsProcessor outputs stored in the fixture."""
    from sensorium_amd.inputs import get_inputs_processor
    for c in range(int(gold["num_cases"])):
        k = _case(gold, c)
        d = k["trials"][0]
        idx = dorc.window_indexes(k["ends"][0], *k["window"])
        proc = get_inputs_processor("stack_inputs", dict(size=k["size"], pad_fill_value=k["fill"]))
        x = proc(d["video"][..., idx], d["behavior"][..., idx], d["pupil_center"][..., idx])
        assert x.dtype.is_floating_point and np.array_equal(x.numpy(), gold[f"c{c}_x0"])
    with pytest.raises(ValueError):
        get_inputs_processor("resize", {})
    with pytest.raises(ValueError):
        get_inputs_processor("stack_inputs", dict(size=(4, 4)))(np.zeros((8, 8, 3)), np.zeros((2, 3)), np.zeros((2, 3)))


def test_oracle_matches_reference_cutmix_draws_and_blend(gold):
    used = 0
    for c in range(int(gold["num_cases"])):
        k = _case(gold, c)
        s = []
        for d, e in zip(k["trials"], k["ends"]):
            idx = dorc.window_indexes(e, *k["window"])
            s.append((dorc.stack_inputs(d["video"][..., idx], d["behavior"][..., idx], d["pupil_center"][..., idx],
                                        k["size"], k["fill"]), dorc.responses_to_target(d["responses"][..., idx])))
        for seed in range(6):
            rs = np.random.RandomState(1000 + seed + 17 * c)
            box = dorc.cutmix_draw(rs, k["size"][1], k["size"][0], 1.0, 0.5)
            assert (box is not None) == bool(gold[f"c{c}_s{seed}_used"])
            if box is None:
                continue
            used += 1
            assert tuple(int(v) for v in gold[f"c{c}_s{seed}_box"]) == box
            x, t = dorc.cutmix_apply(s[0][0], s[0][1], s[1][0], s[1][1], box)
            assert np.array_equal(x, gold[f"c{c}_s{seed}_x"])
            assert np.array_equal(t, gold[f"c{c}_s{seed}_t"])
    assert used >= 8


def test_host_cutmix_box_matches_oracle_draws():
    from sensorium_amd.data_gpu import cutmix_box
    n_used = 0
    for seed in range(200):
        h, w = [(64, 64), (36, 64), (12, 16), (5, 8)][seed % 4]
        a = cutmix_box(np.random.RandomState(seed), h, w, 1.0, 0.5)
        b = dorc.cutmix_draw(np.random.RandomState(seed), h, w, 1.0, 0.5)
        assert a == b
        n_used += a is not None
    assert 60 < n_used < 140


def _samples(k):
    s = []
    for d, e in zip(k["trials"], k["ends"]):
        idx = dorc.window_indexes(e, *k["window"])
        s.append((dorc.stack_inputs(d["video"][..., idx], d["behavior"][..., idx], d["pupil_center"][..., idx],
                                    k["size"], k["fill"]), dorc.responses_to_target(d["responses"][..., idx])))
    return s


def test_oracle_matches_reference_mixup_and_random_choice(golden_dir):
    """Mixup / RandomChoiceMixer([CutMix, Mixup]) of src/mixers.py:22-33,70-79 (tests/golden/data_mixup.npz)."""
    gold = np.load(golden_dir / "data_mixup.npz")
    n_mix = n_box = n_lam = 0
    for c in range(int(gold["num_cases"])):
        k = _case(gold, c)
        s = _samples(k)
        for seed in range(5):
            key = f"c{c}_s{seed}"
            lam = dorc.mixup_draw(np.random.RandomState(2000 + seed + 31 * c), 0.4, 0.7)
            assert (lam is not None) == bool(gold[key + "_mixup_used"])
            if lam is not None:
                assert lam == float(gold[key + "_mixup_lam"])
                x, t = dorc.mixup_apply(s[0][0], s[0][1], s[1][0], s[1][1], lam)
                assert np.array_equal(x, gold[key + "_mixup_x"]) and np.array_equal(t, gold[key + "_mixup_t"])
                n_mix += 1
            rs = np.random.RandomState(3000 + seed + 31 * c)
            assert rs.random_sample() < 1.0                                   # Mixer.use
            which = int(rs.choice(2, p=[0.5, 0.5]))
            assert which == int(gold[key + "_choice"])
            if which == 0:
                box = dorc.cutmix_draw(rs, k["size"][1], k["size"][0], 1.0, None)
                assert box == tuple(int(v) for v in gold[key + "_choice_box"])
                x, t = dorc.cutmix_apply(s[0][0], s[0][1], s[1][0], s[1][1], box)
                n_box += 1
            else:
                lam2 = float(rs.beta(0.4, 0.4))
                assert lam2 == float(gold[key + "_choice_lam"])
                x, t = dorc.mixup_apply(s[0][0], s[0][1], s[1][0], s[1][1], lam2)
                n_lam += 1
            assert np.array_equal(x, gold[key + "_choice_x"]) and np.array_equal(t, gold[key + "_choice_t"])
    assert n_mix >= 5 and n_box >= 2 and n_lam >= 2


def test_host_mixer_draws_follow_the_reference_order():
    """BatchAssembler.draw_train_picks: sample, Mixer.use, partner sample, then the mixer's own draws (datasets.py:121-129)."""
    from sensorium_amd.data_gpu import mixer_call, parse_mixer
    spec = parse_mixer(("random_choice", {"mixers": [("cutmix", {"alpha": 1.0}), ("mixup", {"alpha": 0.4})],
                                          "choice_probs": [0.5, 0.5], "prob": 1.0}))
    n_box = n_lam = 0
    for seed in range(100):
        rs, ro = np.random.RandomState(seed), np.random.RandomState(seed)
        box, lam = mixer_call(rs, spec, 36, 64)
        which = int(ro.choice(2, p=[0.5, 0.5]))
        if which == 0:
            assert lam is None and box == dorc.cutmix_draw(ro, 36, 64, 1.0, None)
            n_box += 1
        else:
            assert box is None and lam == float(ro.beta(0.4, 0.4))
            n_lam += 1
    assert n_box > 25 and n_lam > 25
    assert parse_mixer({"alpha": 1.0, "prob": 0.5}) == dict(kind="cutmix", alpha=1.0, prob=0.5)
    assert parse_mixer(None) is None
    with pytest.raises(ValueError):
        parse_mixer(("blur", {}))


def test_mice_sample_and_collate_structure():
    t = np.arange(12, dtype=np.float32).reshape(3, 4)
    targets, w = dorc.mice_sample(1, t, (2, 3, 5))
    assert [a.shape for a in targets] == [(2, 4), (3, 4), (5, 4)]
    assert np.array_equal(targets[1], t) and not targets[0].any() and not targets[2].any()
    assert w.tolist() == [0.0, 1.0, 0.0]
    x = np.zeros((5, 4, 2, 2), np.float32)
    xb, (tb, wb) = dorc.collate([(x, (targets, w)), (x + 1, dorc.mice_sample(2, np.ones((5, 4), np.float32), (2, 3, 5)))])
    assert xb.shape == (2, 5, 4, 2, 2) and [a.shape for a in tb] == [(2, 2, 4), (2, 3, 4), (2, 5, 4)]
    assert wb.tolist() == [[0, 1, 0], [0, 0, 1]]


def test_window_indexes_positions():
    assert dorc.window_indexes(30, 16, 2) == list(range(0, 31, 2))
    assert dorc.window_indexes(0, 4, 2, "first") == [0, 2, 4, 6]
    assert dorc.window_indexes(10, 5, 1, "middle") == [8, 9, 10, 11, 12]
    from sensorium_amd.predictors import IndexesGenerator
    g = IndexesGenerator(16, 2, "last")
    assert g.make_indexes(30) == dorc.window_indexes(30, 16, 2) and g.behind == 30 and g.width == 31
