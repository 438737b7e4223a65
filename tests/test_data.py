"""Input-layout contract of gen.py:41-67 (what the kernels receive) and the prefetcher."""
import numpy as np

from lstm_ctc_ocr_amd.config import cfg
from lstm_ctc_ocr_amd.utils import gen
from lstm_ctc_ocr_amd.utils.data_util import GeneratorEnqueuer


def test_group_batch_layout():
    imgs = [np.full((60, 160), 200, np.uint8), np.full((60, 120), 100, np.uint8)]
    batch, label_vec, label_len, steps = gen.groupBatch(imgs, ['ab3', 'Z0'])
    assert steps == [85 // 4 - 1, 64 // 4 - 1]                 # nw = int(32/60*w); nw // POOL_SCALE + OFFSET_TIME_STEP
    assert batch[0].shape == (88, 32) and batch[1].shape == (88, 32)      # padded to max width rounded up to x4
    assert batch[0].dtype == np.float32 and abs(batch[0][0, 0] - 200 / 255.) < 1e-6
    assert np.all(batch[1][64:] == 0) and np.all(batch[0][85:] == 0)       # right padding with 0 (black)
    assert label_vec == [11, 12, 4, 62, 1] and label_len == [3, 2]


def test_gray_conversion_quirk_and_generator():
    rgb = np.zeros((1, 3, 3), np.uint8); rgb[0, 0, 0] = 255; rgb[0, 1, 1] = 255; rgb[0, 2, 2] = 255
    assert gen.to_gray_reference(rgb)[0].tolist() == [29, 150, 76]         # BGR weights applied to RGB
    g = gen.generator(batch_size=3)
    imgs, labels, lens, steps = next(g)
    assert len(imgs) == 3 and all(cfg.MIN_LEN <= l <= cfg.MAX_LEN for l in lens) and sum(lens) == len(labels)
    assert imgs[0].shape[1] == 32 and imgs[0].shape[0] % 4 == 0 and all(s == 20 for s in steps)
    assert all(1 <= v <= 62 for v in labels)


def test_enqueuer_threads():
    def counter():
        i = 0
        while True:
            yield i
            i += 1
    e = GeneratorEnqueuer(counter(), use_multiprocessing=False)
    e.start(workers=1, max_queue_size=4)
    got = [e.queue.get(timeout=5) for _ in range(3)]
    assert e.is_running() and got == [0, 1, 2]
    e.stop()
    assert not e.is_running()


def test_data_streams_differ_per_rank_and_per_stream(monkeypatch):
    """Round-1 finding: every data-parallel rank drew the SAME samples (and the validation batch replayed the first training
    batch).  Seeds now depend on $RANK and on the stream kind; two ranks' first batches and train/val batches differ."""
    from lstm_ctc_ocr_amd.utils import gen
    from lstm_ctc_ocr_amd.utils.data_util import GeneratorEnqueuer

    def first_labels(seed):
        e = GeneratorEnqueuer(gen.generator(batch_size=4), use_multiprocessing=False, random_seed=seed)
        e.start(workers=1, max_queue_size=2)
        try:
            return e.queue.get(timeout=60)[1]
        finally:
            e.stop(timeout=5)

    monkeypatch.setenv('RANK', '0')
    s0, v0 = gen.stream_seed(), gen.stream_seed(stream=1)
    monkeypatch.setenv('RANK', '1')
    s1 = gen.stream_seed()
    assert len({s0, s1, v0}) == 3
    assert all(abs(a - b) > 64 for a, b in ((s0, s1), (s0, v0), (s1, v0)))     # workers add their index (< 64) to the seed
    l0, l1, lv = first_labels(s0), first_labels(s1), first_labels(v0)
    assert l0 != l1 and l0 != lv
    assert first_labels(s0) == l0                                               # and each stream is reproducible
