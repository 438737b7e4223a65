"""ORACLE (test infrastructure): executes ANY network plan built with the layer DSL (lstm_ctc_ocr_amd.network) on the CPU
with the same op restatements as oracle/graph.py — used for configurations that are not in the reference (residual
extractors, stacked BiLSTMs), where there is no fixed graph to transcribe.  Differentiable through torch autograd."""
import torch

from . import graph as og


def forward(net, params, x, seq_len, sim_bf16=False, keep=False):
    """net: a lstm_ctc_ocr_amd Network (plan only, never the engine); params: {TF-style name: tensor}."""
    sim = sim_bf16
    cache, inter = {}, {}

    def ev(nd):
        if id(nd) in cache:
            return cache[id(nd)]
        if nd.op == 'input':
            out = x.unsqueeze(3) if nd.name == 'data' else None
        elif nd.op == 'conv':
            a = nd.attrs
            h = ev(nd.inputs[0])
            z = og.conv_single(h, params[nd.name + '/weights'], params[nd.name + '/biases'], a['padding'], sim, first=(a['c_i'] == 1))
            if a['bn']:
                z = og.qa(z, sim)
                z = og.batch_norm_train(z, params['%s/%s/gamma' % (nd.name, nd.name)], params['%s/%s/beta' % (nd.name, nd.name)])
            if a['relu']:
                z = torch.relu(z)
            out = og.qa(z, sim)
        elif nd.op == 'max_pool':
            out = og.qa(og.max_pool(ev(nd.inputs[0]), nd.attrs['k_h'], nd.attrs['k_w']), sim, fwd=False)
        elif nd.op == 'add':
            out = og.qa(ev(nd.inputs[0]) + ev(nd.inputs[1]), sim)
        elif nd.op == 'relu':
            out = og.qa(torch.relu(ev(nd.inputs[0])), sim, fwd=False)
        elif nd.op in ('reshape_squeeze',):
            h = ev(nd.inputs[0])
            out = h.reshape(h.shape[0], h.shape[1] * h.shape[2], h.shape[3])
        elif nd.op == 'dropout':
            out = ev(nd.inputs[0])
        elif nd.op == 'bi_lstm':
            feat = ev(nd.inputs[0])
            fw = og.lstm_direction(feat, seq_len, params[nd.name + '/fw/weights'], params[nd.name + '/fw/biases'], False, sim)
            bw = og.lstm_direction(feat, seq_len, params[nd.name + '/bw/weights'], params[nd.name + '/bw/biases'], True, sim)
            hcat = torch.cat([fw, bw], dim=2)
            if nd.attrs.get('with_fc', True):
                N, T, _ = hcat.shape
                lg = hcat.reshape(N * T, -1) @ og.q(params[nd.name + '/weights'], sim) + params[nd.name + '/biases']
                out = og.qa(lg.reshape(N, T, -1).permute(1, 0, 2).contiguous(), sim, fwd=False)
            else:
                out = hcat
        else:
            raise NotImplementedError(nd.op)
        cache[id(nd)] = out
        if keep:
            inter[nd.name] = out
        return out

    logits = ev(net.get_output('logits'))
    return (logits, inter) if keep else logits
