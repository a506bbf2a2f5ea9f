"""A model of the credit / arrival rendezvous of overlap modes 4 - 6 (wafer_engine_comm.hip, copy_exchange), run under random
schedules on the CPU: no GPU, no engine -- the PROTOCOL.  Every rank is a set of in-order streams of operations exactly as
copy_exchange and launch_halves_pass enqueue them; a scheduler picks any stream whose next operation can run.  Checked:

  * no schedule deadlocks (ranks grant their credits before they wait for anybody's);
  * a copy into a rank's ghost planes never starts before that rank has finished every read of what the planes held before
    (the hook's "the receive is posted" half of the contract), and
  * a read of ghost planes never starts before the copy it is meant to see has landed (the "returns with the planes filled" half),

for the two-sided exchange behind a pass (modes 5 / 6, excited-state steps), for the one-direction exchanges of the single-launch
pass with a stream per side (mode 4; ping-pong buffers), and for refills of ONE buffer (phi changed in place between exchanges:
normalise, observables, ensure_halo), on chains of 1 (its own neighbour: the self-loop of the benches), 2, 3 and 5 ranks.  With
ping-pong buffers the arrivals alone order everything; the credits are what keeps a refill of the same buffer off planes still
being read -- the last test removes them and finds the schedule that breaks.  The words are the engine's: per rank CREDIT[side], ARRIVED[side], written by that side's neighbour only; the
counts are per link and direction and only grow."""
import random

import pytest


class Rank:
    def __init__(self, r, world, self_loop=False):
        self.r = r
        self.lo = r if self_loop else (r - 1 if r > 0 else None)
        self.hi = r if self_loop else (r + 1 if r + 1 < world else None)
        self.credit = [0, 0]       # CREDIT[side]: receives the neighbour on that side has posted for what I send it
        self.arrived = [0, 0]      # ARRIVED[side]: copies of that side's neighbour that have landed in my ghost planes
        # ghost planes of a side, per ping-pong buffer: how often they were filled / read, and whether a copy is writing them now.
        # Buffer 0 starts filled once (the initial condition is generated with its ghost planes).
        self.fills = {(s, b): (1 if b == 0 else 0) for s in (0, 1) for b in (0, 1)}
        self.reads = {(s, b): 0 for s in (0, 1) for b in (0, 1)}
        self.writing = {(s, b): False for s in (0, 1) for b in (0, 1)}
        self.sent = [0, 0]
        self.recv = [0, 0]
        self.streams = []          # lists of ops
        self.half_done = {}        # (pass, half) -> True once that half's workgroups have finished

    def nb(self, side):
        return self.lo if side == 0 else self.hi


def exchange_ops(rank, sides_send, sides_recv, buf, refill=False):
    """what copy_exchange enqueues for the planes of ping-pong buffer `buf`: PRE (grant my receives, then wait for the credits of my
    sends), the copies, POST (announce my copies, then wait for the arrivals of my receives).  A side without a neighbour takes no
    part.  refill: the planes are filled again without having been read (ensure_halo on a buffer whose planes are current)."""
    send = [s for s in sides_send if rank.nb(s) is not None]
    recv = [s for s in sides_recv if rank.nb(s) is not None]
    if not send and not recv:
        return []
    grants = []
    for s in recv:
        rank.recv[s] += 1
        grants.append((s, rank.recv[s]))
    waits = [(s, rank.sent[s] + 1) for s in send]
    ops = [("pre", grants, waits)]
    announces = []
    for s in send:
        rank.sent[s] += 1
        ops.append(("copy_begin", s, buf, refill))
        ops.append(("copy_end", s, buf))
        announces.append((s, rank.sent[s]))
    ops.append(("post", announces, [(s, rank.recv[s]) for s in recv]))
    return ops


def run(ranks, rng, max_steps=400000):
    """random scheduler; returns the number of operations executed.  Raises on a safety violation, asserts on deadlock."""
    pcs = {(k.r, i): 0 for k in ranks for i in range(len(k.streams))}
    by_r = {k.r: k for k in ranks}
    pending_wait = {}          # (r, stream): the stores of a pre / post are done, its waits are not
    executed = 0
    for _ in range(max_steps):
        runnable = []
        for (r, i), pc in pcs.items():
            k = by_r[r]
            if pc >= len(k.streams[i]):
                continue
            op = k.streams[i][pc]
            if op[0] in ("pre", "post"):
                word = k.credit if op[0] == "pre" else k.arrived
                if (r, i) not in pending_wait or all(word[s] >= v for s, v in op[2]):
                    runnable.append((r, i))
            elif op[0] == "gate":          # the exchange stream waits for a half of a pass
                if k.half_done.get(op[1]):
                    runnable.append((r, i))
            elif op[0] == "read":          # a kernel (or a half's boundary workgroups) about to read the ghost planes of (side, buffer)
                key = (op[1], op[2])
                ready = k.nb(op[1]) is None or (k.fills[key] == k.reads[key] + 1 and not k.writing[key])
                if ready:
                    runnable.append((r, i))
                elif op[3] == "stream_order":   # the read sits behind the exchange in its stream: it must be satisfied by construction
                    raise AssertionError(f"rank {r}: a read of {key} behind its exchange finds fills {k.fills[key]}, reads {k.reads[key]}, writing {k.writing[key]}")
                # ("flag": the workgroups poll a flag the side's chain posts behind the arrival: they simply wait)
            else:
                runnable.append((r, i))
        if not runnable:
            assert all(pc >= len(by_r[r].streams[i]) for (r, i), pc in pcs.items()), \
                f"deadlock: {[(r, i, by_r[r].streams[i][pc]) for (r, i), pc in pcs.items() if pc < len(by_r[r].streams[i])]}"
            return executed
        r, i = rng.choice(runnable)
        k = by_r[r]
        op = k.streams[i][pcs[(r, i)]]
        if op[0] in ("pre", "post"):
            if (r, i) not in pending_wait:      # the stores: never wait
                for s, v in op[1]:
                    n, back = by_r[k.nb(s)], 1 - s      # the neighbour knows me as its neighbour on the other side
                    word = n.credit if op[0] == "pre" else n.arrived
                    assert word[back] == v - 1, "the counts of a link and direction grow by one"
                    word[back] = v
                pending_wait[(r, i)] = True
                word = k.credit if op[0] == "pre" else k.arrived
                if not all(word[s] >= v for s, v in op[2]):
                    continue                    # the waits are not satisfied yet: the kernel keeps spinning
            del pending_wait[(r, i)]
        elif op[0] == "copy_begin":
            s, buf, refill = op[1], op[2], op[3]
            n, key = by_r[k.nb(s)], (1 - s, op[2])
            # the receiver must have read what these planes held (every earlier fill), and nobody else may be writing them
            if n.reads[key] < n.fills[key] and not refill:
                raise AssertionError(f"a copy of rank {r} into rank {n.r}'s ghost planes {key} while fill {n.fills[key]} has not been read")
            if n.writing[key]:
                raise AssertionError(f"two copies into rank {n.r}'s ghost planes {key} at once")
            n.writing[key] = True
            if refill and n.reads[key] < n.fills[key]:
                n.fills[key] -= 1               # the unread fill is replaced
        elif op[0] == "copy_end":
            n, key = by_r[k.nb(op[1])], (1 - op[1], op[2])
            n.writing[key] = False
            n.fills[key] += 1
        elif op[0] == "read":
            key = (op[1], op[2])
            if k.nb(op[1]) is not None:
                k.reads[key] += 1
        elif op[0] == "half":
            k.half_done[op[1]] = True
        pcs[(r, i)] += 1
        executed += 1
    raise AssertionError("the model did not finish")


@pytest.mark.parametrize("world,self_loop", [(1, True), (2, False), (3, False), (5, False)])
@pytest.mark.parametrize("seed", range(6))
def test_two_sided_exchange_behind_every_pass(world, self_loop, seed):
    """modes 5 / 6, excited-state steps: per pass ONE stream does [kernel reads both ghost sides of the input buffer] [exchange of both
    sides of the output buffer]; the buffers ping-pong."""
    rng = random.Random(seed)
    ranks = [Rank(r, world, self_loop) for r in range(world)]
    passes = 7
    for k in ranks:
        ops = []
        for p in range(passes):
            ops += [("read", 0, p & 1, "stream_order"), ("read", 1, p & 1, "stream_order")]
            ops += exchange_ops(k, (0, 1), (0, 1), (p + 1) & 1)
        k.streams = [ops]
    n = run(ranks, rng)
    assert n == sum(len(k.streams[0]) for k in ranks)
    for k in ranks:
        for side in (0, 1):
            if k.nb(side) is not None:
                assert k.credit[side] == passes and k.arrived[side] == passes


@pytest.mark.parametrize("world,self_loop", [(1, True), (2, False), (3, False), (5, False)])
@pytest.mark.parametrize("seed", range(6))
def test_one_direction_exchanges_of_the_single_launch_pass_with_a_stream_per_side(world, self_loop, seed):
    """mode 4: the pass runs on the main stream as two halves; half A (lower) reads the LOWER ghost planes of the input buffer late
    and half B the UPPER ones, each behind its side's flag (modelled as: the read waits until the planes hold a fill nobody has
    read -- the workgroups poll); when a half is done its side's chain -- gate, exchange of ONE direction, flag -- runs on that
    side's own stream: side 0 sends my lowest planes DOWN and receives into the UPPER ghost planes of the output buffer, side 1 the
    mirror image.  The order of the halves alternates per pass.  The first pass's ghost planes come from a two-sided exchange on
    the main stream (ensure_halo: a refill of planes that nobody has read since they were generated)."""
    rng = random.Random(100 + seed)
    ranks = [Rank(r, world, self_loop) for r in range(world)]
    passes = 6
    for k in ranks:
        main, aux = [], [[], []]
        main += exchange_ops(k, (0, 1), (0, 1), 0, refill=True)
        for p in range(passes):
            first = p & 1
            for i in range(2):
                half = (first + i) & 1
                main.append(("read", half, p & 1, "flag"))          # half A reads side 0, half B side 1, of the input buffer
                main.append(("half", (p, half)))
            for half in (0, 1):
                aux[half].append(("gate", (p, half)))
                aux[half] += exchange_ops(k, (half,), (1 - half,), (p + 1) & 1)   # send on my side `half`, receive into the OTHER ghost side
        k.streams = [main, aux[0], aux[1]]
    run(ranks, rng)
    for k in ranks:
        for side in (0, 1):
            if k.nb(side) is not None:
                assert k.credit[side] == passes + 1 and k.arrived[side] == passes + 1


def same_buffer_program(k, steps, with_credits=True):
    """phi changed in place between exchanges (normalise, Gram-Schmidt, a download in between): the ghost planes of ONE buffer are
    read and refilled over and over -- the case in which nothing but the credit keeps a copy off planes that are still being read
    (with ping-pong buffers the arrivals alone order everything: the two tests above pass without the credit waits too)"""
    ops = []
    for _ in range(steps):
        ops.append(("read", 0, 0, "stream_order"))
        ops.append(("read", 1, 0, "stream_order"))
        ex = exchange_ops(k, (0, 1), (0, 1), 0)
        if ex and not with_credits:
            ex = [("pre", ex[0][1], [])] + ex[1:]           # grants kept, the waits for credit removed
        ops += ex
    return ops


@pytest.mark.parametrize("world,self_loop", [(1, True), (2, False), (3, False), (5, False)])
@pytest.mark.parametrize("seed", range(6))
def test_refills_of_one_buffer_are_kept_off_planes_still_being_read(world, self_loop, seed):
    rng = random.Random(200 + seed)
    ranks = [Rank(r, world, self_loop) for r in range(world)]
    for k in ranks:
        k.streams = [same_buffer_program(k, 8)]
    run(ranks, rng)


def test_the_model_catches_a_protocol_without_credits():
    """the check is not vacuous: drop the credit wait (copy as soon as the sender is ready) and some schedule writes a rank's ghost
    planes before it has read the old ones"""
    caught = 0
    for seed in range(60):
        rng = random.Random(seed)
        ranks = [Rank(r, 3) for r in range(3)]
        for k in ranks:
            k.streams = [same_buffer_program(k, 6, with_credits=False)]
        try:
            run(ranks, rng)
        except AssertionError as e:
            if "has not been read" in str(e) or "at once" in str(e) or "behind its exchange" in str(e):
                caught += 1
    assert caught > 0
