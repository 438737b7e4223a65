"""Sequence (exact-match) accuracy — /root/reference/lib/lstm/utils/training.py:26-37, the metric behind the README's
">95 %" claim.  Zeros (the CTC blank / dense padding value) are dropped from both sides before comparing."""
from ..config import cfg


def accuracy_calculation(original_seq, decoded_seq, ignore_value=0, isPrint=True):
    if len(original_seq) != len(decoded_seq):
        print('original lengths is different from the decoded_seq,please check again')
        return 0
    count = 0
    for i, origin_label in enumerate(original_seq):
        decoded_label = [j for j in decoded_seq[i] if j != ignore_value]
        org_label = [l for l in origin_label if l != ignore_value]
        if isPrint and i < cfg.VAL.PRINT_NUM:
            print('seq{0:4d}: origin: {1} decoded:{2}'.format(i, origin_label, decoded_label))
        if org_label == decoded_label:
            count += 1
    return count * 1.0 / len(original_seq)
