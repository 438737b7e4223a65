"""ORACLE — test infrastructure, never the product path.

CPU restatement of the CRNN-OCR hot path of ilovin/lstm_ctc_ocr (TF-1.0.1 + baidu warp-ctc semantics,
SURVEY.md Appendix A).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; nothing under lstm_ctc_ocr_amd/ does.

Parity status: the reference holds NO golden vectors or tests for this path and its arithmetic lives in
un-vendored third-party code (tensorflow 1.0.1, warp-ctc master), none of which is installable here.  The
oracle is therefore pinned on (a) warp-ctc's published known-answer vector, (b) brute-force path enumeration,
(c) torch.nn.functional.ctc_loss / torch CPU conv, pool, batch-norm kernels as independent implementations,
(d) a hand-written numpy LSTM cross-checked against a re-packed torch.nn.LSTM, and (e) the vectors TensorFlow's own unit tests hold
for the ops of this graph: ctc_loss_op_test testBasic (3.34211 / 5.42262), ctc_decoder_ops_test testCTCDecoderBeamSearch,
core_rnn_cell_test testBasicLSTMCell (cell equations, forget_bias), conv_ops_test testConv2D2x2Filter / 1x1Filter, max-pool VALID,
clip_ops_test testClipByGlobalNormClipped.  Still unpinned against TensorFlow: contrib batch-norm, Adam, the composed graph.
See tests/test_oracle_*.py.
"""
