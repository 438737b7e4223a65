from lstm_ctc_ocr_amd.utils.gen import *  # noqa: F401,F403
from lstm_ctc_ocr_amd.utils.gen import get_batch, generator, groupBatch, gen_rand, generateImg  # noqa: F401
