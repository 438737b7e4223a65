from lstm_ctc_ocr_amd.config import *  # noqa: F401,F403
from lstm_ctc_ocr_amd.config import cfg, cfg_from_file, cfg_from_list, get_output_dir, get_log_dir, get_encode_decode_dict, _merge_a_into_b  # noqa: F401
