"""Host-side pieces of the training step that run without a GPU: the CutMix rectangles the trainer draws must be the
oracle's maps (train.py:222-251) under the same RNG streams, draw for draw."""
import random

import numpy as np

from oracle import host_ref as H


def test_cutmix_rectangles_follow_the_reference_draw_order():
    from ustrun import trainer as T
    for seed in range(6):
        random.seed(seed); np.random.seed(seed)
        ref = [H.cutmix_box(96, p=0.6) for _ in range(12)]
        tail_ref = (random.random(), np.random.rand())
        random.seed(seed); np.random.seed(seed)
        rects = [T.cutmix_rect(96, p=0.6) for _ in range(12)]
        tail = (random.random(), np.random.rand())
        assert tail == tail_ref                                    # both RNG streams advanced identically
        for r, m in zip(rects, ref):
            assert np.array_equal(T.rect_map(r, 96), m)
        random.seed(seed); np.random.seed(seed)
        assert all(np.array_equal(T.cutmix_box(96, p=0.6), m) for m in ref)


def test_all_cover_rectangle():
    from ustrun import trainer as T
    rng = np.random.default_rng(3)
    for _ in range(20):
        region = (rng.random((40, 40)) > 0.97).astype(np.float32)
        if region.sum() == 0:
            continue
        assert np.array_equal(T.all_cover_box(region), H.all_cover_box(region))
    random.seed(1); np.random.seed(1)
    ref = H.all_cover_box(np.zeros((40, 40), dtype=np.float32))
    random.seed(1); np.random.seed(1)
    assert np.array_equal(T.all_cover_box(np.zeros((40, 40), dtype=np.float32)), ref)
