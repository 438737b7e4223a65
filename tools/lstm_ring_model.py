"""CPU model of the persistent LSTM kernels' inter-workgroup hand-off (lstm_seq.hip, protocol 4: data-as-flag through a ring of RING = 4 step
slots in one XCD's L2) — the hardware-free counterpart of tools/lstm_tail_race_probe.py, explored EXHAUSTIVELY over interleavings.

What the kernels do, per workgroup w of a (direction, batch tile) group and iteration i = 0 .. T - 1 (reference: the tf.while_loop of
lib/networks/network.py:104-109 that these kernels replace):
    poll     i > 0: re-read every producer's piece of slot (i - 1) & 3 until no element of a row that NEEDS it holds the fill pattern
             (a row needs the hand-off of iteration i iff it is inside its sequence at i - 1 AND at i: `active` in the forward kernels, `has_next`
             in the backward ones); rows that do not need it read a don't-care value.  In the four-wave kernels every wave polls a quarter
             of the producers; the waves meet at a barrier before anything is stored.
    payload  store the own piece of slot i & 3 (rows outside their sequence store zeros)
    refill   i >= 2: store the fill pattern over the own piece of slot (i - 2) & 3
Stores are plain and are NOT drained: they land some time after issue, in issue order (vmcnt retires in order); the rows (cache lines) of one
store instruction land in any order; the storing wave's next successful poll covers them (its wait is a wait for every older vector-memory
operation of the wave).  Only producer p ever writes p's pieces.

    rule 'always'   (rounds 3-4, build 7a5d3057 and before): payload and refill are issued in EVERY iteration.
    rule 'live'     (round 5, 01939a7 forward / 9fec851 backward): a workgroup none of whose rows is inside its sequence at i leaves the
                    ring alone in that iteration — it polls nothing there, so it runs free, and its stores would land in slots a slower
                    workgroup is still reading.

Checked in every reachable state, for W workgroups x R rows x every assignment of sequence lengths:
    stale     a poll that passes takes, for a row that needs it, anything but the payload producer p stored for that row at iteration i - 1
              (an older payload that was never refilled, the zero payload of a free iteration, a later one): the silently wrong h / dz
    overwrite a store lands on a piece whose current payload some workgroup has not read yet for a row that needs it
    timeout   a workgroup waits for a piece that will never arrive (all runnable work done, somebody still blocked): the expired bounded spin

The memory image is a function of how many store parts of each producer have landed (single writer per piece, in-order landing), so a global
state is just (program counter, passed sub-polls, landed parts) per workgroup and the exploration is a plain depth-first search.

    python tools/lstm_ring_model.py            # prints the table tests/test_lstm_ring_model.py asserts
"""
import itertools
import sys

RING = 4
FILL = -1


class Violation(Exception):
    def __init__(self, kind, detail):
        super().__init__('%s: %s' % (kind, detail))
        self.kind = kind


def activity(direction, T, lens):
    """act[r][i]: row r is inside its sequence at iteration i.  forward kernels: step s = i, active = s < len (both the fw and the bw direction of
    the layer walk s upwards: lstm_seq.hip `active`); backward kernels: s = T - 1 - i, active = s < len — their free iterations come FIRST."""
    if direction == 'fwd':
        return [[i < L for i in range(T)] for L in lens]
    return [[(T - 1 - i) < L for i in range(T)] for L in lens]


def build_program(rule, act, T):
    """Instruction list of ONE workgroup (all workgroups of a group share the batch tile, hence the program): ('poll', i) / ('store', slot, tag)
    with tag = iteration of the payload or FILL."""
    R = len(act)
    prog = []
    for i in range(T):
        if i > 0:
            prog.append(('poll', i))
        live = True if rule == 'always' else any(act[r][i] for r in range(R))
        if live:
            prog.append(('store', i & (RING - 1), i))
            if i >= 2:
                prog.append(('store', (i - 2) & (RING - 1), FILL))
    return prog


def explore(rule, direction, T, lens, W=3, waves=1, max_states=2_000_000):
    """Exhaustive search over interleavings.  Returns (#states, None) or raises Violation.  waves: 1 (one polling wave per workgroup: the
    lstm_*_seq_kernel family) or >1 (producer p's piece is polled by wave p % waves; only wave 0 issues the ring stores, so only ITS sub-polls
    wait for the workgroup's older stores to land: the lstm_*_seq4* family)."""
    R = len(lens)
    act = activity(direction, T, lens)
    needs = [[i > 0 and act[r][i] and act[r][i - 1] for i in range(T)] for r in range(R)]
    prog = build_program(rule, act, T)
    stores = [k for k, ins in enumerate(prog) if ins[0] == 'store']          # program indices of the store instructions
    nstores_before = [sum(1 for k in stores if k < pc) for pc in range(len(prog) + 1)]
    full = (1 << R) - 1
    allp = (1 << W) - 1

    def memory(landed, headmask):
        """piece image of ONE producer: mem[slot][row] = tag, from `landed` complete store instructions + the rows in headmask of the next one"""
        mem = [[FILL] * R for _ in range(RING)]
        for n in range(landed):
            _, slot, tag = prog[stores[n]]
            for r in range(R):
                mem[slot][r] = tag
        if headmask:
            _, slot, tag = prog[stores[landed]]
            for r in range(R):
                if headmask >> r & 1:
                    mem[slot][r] = tag
        return mem

    mem_cache = {}

    def mem_of(landed, headmask):
        key = (landed, headmask)
        m = mem_cache.get(key)
        if m is None:
            m = mem_cache[key] = memory(landed, headmask)
        return m

    poll_pc = {ins[1]: k for k, ins in enumerate(prog) if ins[0] == 'poll'}
    norm_cache = {}

    def norm1(wst):
        """run the local, invisible actions of ONE workgroup eagerly (sound partial-order reduction): issuing a store only appends to the
        workgroup's own queue; a sub-poll none of whose rows needs the hand-off passes without looking"""
        out = norm_cache.get(wst)
        if out is not None:
            return out
        pc, polled, landed, head = wst
        while pc < len(prog):
            ins = prog[pc]
            if ins[0] == 'store':
                pc += 1
                continue
            if polled == allp or not any(needs[r][ins[1]] for r in range(R)):
                pc += 1
                polled = 0
                continue
            break
        out = norm_cache[wst] = (pc, polled, landed, head)
        return out

    w0 = norm1((0, 0, 0, 0))
    start = tuple(w0 for _ in range(W))
    seen = {start}
    stack = [start]
    nprog = len(prog)
    while stack:
        st = stack.pop()
        progress = False
        for w in range(W):
            pc, polled, landed, head = st[w]
            issued = nstores_before[pc]
            # (1) one more row of the oldest store in flight lands
            if landed < issued:
                _, slot, tag = prog[stores[landed]]
                cur = mem_of(landed, head)[slot]
                for r in range(R):
                    if head >> r & 1:
                        continue
                    old = cur[r]
                    if old != FILL and old != tag:
                        # a payload is being overwritten: has everybody who needs it read it?  (payload of iteration `old` is polled at old + 1)
                        i = old + 1
                        if i < T and needs[r][i]:
                            k = poll_pc[i]
                            for q in range(W):
                                qpc, qpolled = st[q][0], st[q][1]
                                if qpc < k or (qpc == k and not (qpolled >> w & 1)):
                                    raise Violation('overwrite', 'rule %s %s T=%d lens=%s: workgroup %d overwrites its piece of slot %d (payload of iteration %d, '
                                                    'row %d) with %s before workgroup %d has polled it' % (rule, direction, T, lens, w, slot, old, r,
                                                                                                         'the fill pattern' if tag == FILL else 'iteration %d' % tag, q))
                    nh = head | (1 << r)
                    nxt = norm1((pc, polled, landed + 1, 0) if nh == full else (pc, polled, landed, nh))
                    progress = True
                    s2 = st[:w] + (nxt,) + st[w + 1:]
                    if s2 not in seen:
                        seen.add(s2)
                        stack.append(s2)
            # (2) a sub-poll passes
            if pc < nprog:
                i = prog[pc][1]                           # (a normalised workgroup that is not finished stands at a poll)
                slot = (i - 1) & (RING - 1)
                for p in range(W):
                    if polled >> p & 1:
                        continue
                    if landed < issued and p % waves == 0:
                        continue                          # the storing wave's poll returns behind its own older stores
                    sp = st[p]
                    pm = mem_of(sp[2], sp[3])[slot]
                    ok = True
                    for r in range(R):
                        if needs[r][i] and pm[r] == FILL:
                            ok = False                    # keeps polling
                            break
                    if not ok:
                        continue
                    for r in range(R):
                        if needs[r][i] and pm[r] != i - 1:
                            raise Violation('stale', 'rule %s %s T=%d lens=%s: workgroup %d at iteration %d takes the payload of iteration %d from workgroup %d '
                                            '(row %d) for that of iteration %d' % (rule, direction, T, lens, w, i, pm[r], p, r, i - 1))
                    progress = True
                    s2 = st[:w] + (norm1((pc, polled | (1 << p), landed, head)),) + st[w + 1:]
                    if s2 not in seen:
                        seen.add(s2)
                        stack.append(s2)
        if not progress:
            blocked = [w for w in range(W) if st[w][0] < nprog]
            if blocked:
                w = blocked[0]
                raise Violation('timeout', 'rule %s %s T=%d lens=%s: workgroup %d waits at iteration %d for a piece that never arrives (everything else '
                                'has run to completion)' % (rule, direction, T, lens, w, prog[st[w][0]][1]))
        if len(seen) > max_states:
            raise RuntimeError('state budget exceeded')
    return len(seen)


def sweep(rule, direction, T, W=3, R=2, waves=1, min_len=0):
    """every assignment of sequence lengths min_len .. T to the R rows of the tile -> (patterns, states, {kind: [first message, count]})"""
    found = {}
    states = 0
    npat = 0
    for lens in itertools.product(range(min_len, T + 1), repeat=R):
        npat += 1
        try:
            states += explore(rule, direction, T, lens, W=W, waves=waves)
        except Violation as v:
            ent = found.setdefault(v.kind, [str(v), 0])
            ent[1] += 1
    return npat, states, found


def main():
    for T in (4, 5, 6):
        for direction in ('fwd', 'bwd'):
            for waves in (1, 4):
                for rule in ('always', 'live'):
                    npat, states, found = sweep(rule, direction, T, waves=waves)
                    print('T=%d %s waves=%d rule=%-6s: %3d length patterns, %8d states explored, violations: %s'
                          % (T, direction, waves, rule, npat, states, {k: v[1] for k, v in found.items()} or 'none'))
                    for k, v in found.items():
                        print('      e.g. ' + v[0])
    return 0


if __name__ == '__main__':
    sys.exit(main())
