"""The host planner behind tri_batch_create (csrc/planner.hpp) on a CPU-only machine, through libtrinity_host.so (csrc/host/plan_host.cpp):
what every kernel relies on without checking — a query's tasks are consecutive and cover its windows / tiles exactly once, output regions
never overlap and stay inside the batch's capacity, the schedule is a permutation grouped by kernel with the heavier tasks first, every
term position that names a plane row names the right term — and that the plan does not depend on how many host threads made it."""
import numpy as np
import pytest

import trinity_amd as T
from trinity_amd import hostplan as HP
from trinity_amd import workloads as W


@pytest.fixture(scope="module")
def world():
    T.build.build_host()
    D, V = 200_000, 20_000
    segs = {c: T.Segment(D, V, 10, 42, codec=c) for c in (T.engine.CODEC_GOOGLE, T.engine.CODEC_LUCENE)}
    return D, V, segs, {c: HP.HostIndex.from_segment(s) for c, s in segs.items()}


OPTION_SETS = ({}, {"dense_min_postings": 0}, {"planes": 0}, {"plane_div": 1 << 30, "dense_min_postings": 0}, {"dense_min_postings": 0, "planes_split": 5},
               {"dense_min_postings": 0, "planes_split": 1 << 20, "fused_task_cost": 4096}, {"dense_task_cost": 1000, "dense_min_postings": 0},
               {"plane_div": 1 << 30, "plane_max_bytes": 300_000})  # fmt: skip


def check_plan(p, nq):
    s = p.s
    plan, tasks, sched = p.plan, p.tasks, p.sched
    # queries: plan order = query order; every lowered query knows its slot
    # (a hidden phrase query — a phrase leaf of a TASK_TREE query — has a slot and tasks, and no caller query: qid = 0xffffffff)
    vis = plan["qid"] != 0xFFFFFFFF
    assert np.all(np.diff(plan["qid"][vis].astype(np.int64)) > 0)
    lowered = p.slot_of_query[:nq] != 0xFFFFFFFF
    assert int(lowered.sum()) == int(vis.sum()) and int((~vis).sum()) == s["n_tree_hidden"]
    assert np.array_equal(plan["qid"][vis], np.nonzero(lowered)[0]) and np.array_equal(p.slot_of_query[:nq][lowered], np.nonzero(vis)[0])
    assert np.array_equal(np.sort(p.tree_hidden), np.nonzero(~vis)[0])
    # tasks: consecutive per query, in plan order, covering [0, n) of the query's windows / tiles without gaps
    assert np.array_equal(plan["first_task"].astype(np.int64), np.concatenate([[0], np.cumsum(plan["ntasks"].astype(np.int64))[:-1]]))
    assert int(plan["ntasks"].sum()) == s["n_tasks"] and np.all(plan["ntasks"] >= 1)
    assert np.array_equal(tasks["slot"], np.repeat(np.arange(s["n_plan"], dtype=np.uint32), plan["ntasks"]))
    first = np.zeros(s["n_tasks"], dtype=bool)
    first[plan["first_task"]] = True
    assert np.all(tasks["begin"][first] == 0) and np.all(tasks["begin"] < tasks["end"])
    assert np.all(tasks["begin"][~first] == tasks["end"][:-1][~first[1:]])
    # one kind per query
    kind_q = tasks["kind"][plan["first_task"]]
    assert np.array_equal(tasks["kind"], np.repeat(kind_q, plan["ntasks"]))
    # output regions: the queries' regions tile [0, out_capacity); a task's region starts inside its query's and tasks ascend
    assert np.array_equal(plan["out_off"], np.concatenate([[0], np.cumsum(plan["out_cap"].astype(np.uint64))[:-1]]).astype(np.uint64))
    assert int(plan["out_off"][-1] + plan["out_cap"][-1]) == s["out_capacity"] if s["n_plan"] else s["out_capacity"] == 0
    q_off, q_cap = np.repeat(plan["out_off"], plan["ntasks"]), np.repeat(plan["out_cap"].astype(np.uint64), plan["ntasks"])
    assert np.all(tasks["out_off"] >= q_off) and np.all((tasks["out_off"] <= q_off + q_cap))
    same_q = tasks["slot"][1:] == tasks["slot"][:-1]
    assert np.all(tasks["out_off"][1:][same_q] >= tasks["out_off"][:-1][same_q])
    # candidate tiles: a task's region is exactly its tiles' candidates
    cand = tasks["kind"] == HP.TASK_CAND
    assert np.array_equal(tasks["out_off"][cand], q_off[cand] + tasks["begin"][cand].astype(np.uint64) * 8192)
    # schedule: a permutation, grouped by kernel in launch order, the per-kernel counts as reported
    assert np.array_equal(np.sort(sched), np.arange(s["n_tasks"], dtype=np.uint32))
    counts = [s["n_dense"], s["n_pset"], s["n_probe"], s["n_cand"], s["n_fused"], s["n_fused16"], s["n_fusedgen"], s["n_planes"], s["n_planes8"], s["n_tree"]]
    assert sum(counts) == s["n_tasks"]
    at = 0
    for kind, c in zip(HP.SCHED_ORDER, counts):
        assert np.all(tasks["kind"][sched[at : at + c]] == kind)
        at += c
    assert s["dense_queries"] + s["pset_queries"] + s["probe_queries"] + s["cand_queries"] + s["fused_queries"] + s["planes_queries"] + s["tree_queries"] == s["n_plan"]
    # TASK_TREE: one task per query; a record of postfix nodes — children before parents, one root, every leaf with a row of its own kind;
    # a phrase leaf names a hidden query that precedes the tree query in the plan
    assert s["n_tree"] == s["tree_queries"] and np.all(np.diff(p.tree_terms.astype(np.int64)) > 0)
    for sl in np.nonzero(kind_q == HP.TASK_TREE)[0][:300]:
        assert plan["ntasks"][sl] == 1
        nd = p.tree_nodes(sl)
        assert 1 <= len(nd) <= 64 and nd["parent"][-1] == 0xFF and np.all(nd["parent"][:-1] > np.arange(len(nd) - 1))
        for k, x in enumerate(nd):
            if x["op"] == 0:  # TERM
                assert p.tree_terms[x["row"]] == x["arg"]
            elif x["op"] == 3:  # PHRASE
                assert x["row"] >> 31 and p.tree_hidden[x["row"] & 0x7FFFFFFF] == x["arg"] < sl and not vis[x["arg"]] and plan["nphrases"][x["arg"]] == 1
            else:
                kids = [j for j in range(k) if (int(x["kids"]) >> j) & 1]
                assert kids and all(nd["parent"][j] == k for j in kids) and sorted(nd["ord"][kids].tolist()) == list(range(len(kids)))
    # unit records (k_psets / k_probe): one per task of those kinds in run order, the task's own geometry, every probed term's row inline
    if s["n_pset"] + s["n_probe"]:
        units, us = p.units, p.unit_sched
        assert s["sizeof_unit"] == HP.DEV_UNIT.itemsize and len(set(us.tolist())) == len(us)
        ran = units[us]
        assert np.array_equal(ran["tix"], sched[s["n_dense"] : s["n_dense"] + len(us)])
        tk = tasks[ran["tix"]]
        assert np.array_equal(tk["begin"], ran["begin"]) and np.array_equal(tk["end"], ran["end"]) and np.array_equal(tk["out_off"], ran["out_off"])
        assert np.all(tk["kind"][: s["n_pset"]] == HP.TASK_PSET) and np.all(tk["kind"][s["n_pset"] :] == HP.TASK_PROBE)
        # plane-set tasks run window range (PSET_TASK_WINDOWS = 4 windows) by window range — a query with phrases may cut its range finer (phrase_task_div)
        assert np.all(np.diff(ran["begin"][: s["n_pset"]].astype(np.int64) // 4) >= 0)
        for u in ran[:: max(1, len(ran) // 300)]:
            k0 = 1 if tasks[u["tix"]]["kind"] == HP.TASK_PROBE else 0
            for k in range(k0, min(int(u["nterms"]), 4)):
                scatter = bool(u["first"] & 4)  # PSET_UNIT_SCATTER: a union may name terms without a plane (their documents are set in the stored words)
                assert (u["tt"][k] & 0x3FFFFFFF) == (p.qterms[u["term_base"] + k] & 0x3FFFFFFF) and u["row"][k] == p.qplane[u["term_base"] + k]
                assert scatter or u["row"][k] != 0xFFFFFFFF
            if u["first"] & 4:  # ... but at least one of its terms has one, it is ONE group with nothing excluded, and its result is a bitmap
                tw = p.qterms[u["term_base"] : u["term_base"] + int(u["nterms"])]
                assert (u["first"] & 2) and (tw[0] >> 31) and not np.any(tw[1:] >> 31) and not np.any((tw >> 30) & 1)
                assert np.any(p.qplane[u["term_base"] : u["term_base"] + int(u["nterms"])] != 0xFFFFFFFF)
    # planes: a term position that names a row names its own term's row
    if s["n_qplane"]:
        # (a row is the term's rank by document count — the planes live with the index, every batch names the same row for the same term)
        qt, qp, rows = p.qterms & 0x3FFFFFFF, p.qplane, p.plane_terms
        named = qp != 0xFFFFFFFF
        assert np.all(np.isin(qt[named], rows)) and np.all(np.diff(rows.astype(np.int64)) > 0)
        row_of = {}
        for t_, r_ in zip(qt[named].tolist(), qp[named].tolist()):
            assert row_of.setdefault(t_, r_) == r_  # one row per term ...
        assert len(set(row_of.values())) == len(row_of)  # ... and one term per row


@pytest.mark.parametrize("wl", ["cfg2", "cfg3", "cfg4", "cfg5"])
def test_plan_invariants_and_thread_independence(world, wl):
    D, V, segs, hix = world
    nq = 3000
    parts, _ = W.build_parts(wl, D, V, 10, 42, nq)
    for pt in parts:
        for opts in OPTION_SETS:
            p1 = HP.HostPlan(hix[pt.codec], pt.programs, pt.flags, pt.topk, threads=1, options=opts)
            check_plan(p1, len(pt.programs))
            for th in (3, 8):
                pn = HP.HostPlan(hix[pt.codec], pt.programs, pt.flags, pt.topk, threads=th, options=opts)
                assert pn.s == p1.s, opts
                assert np.array_equal(pn.block, p1.block), (wl, opts, th)  # the same bytes go to the device whatever the thread count
                assert np.array_equal(pn.slot_of_query, p1.slot_of_query) and np.array_equal(pn.qstatus, p1.qstatus)
                pn.close()
            p1.close()


def test_heavier_tasks_are_scheduled_first(world):
    D, V, segs, hix = world
    parts, _ = W.build_parts("cfg2", D, V, 10, 42, 4000)
    p = HP.HostPlan(hix[parts[0].codec], parts[0].programs, parts[0].flags, 0, threads=4, options={"dense_min_postings": 100_000})
    tasks, sched = p.tasks, p.sched
    # candidate-tile tasks of a 2-term AND: cost = tiles x (lead postings + 32 x min(other blocks, lead documents)) / tiles; the schedule's
    # order is by cost octave + 3 bits, so within a kernel the spans never grow by more than one bucket (12.5 %) from one task to the next
    span = (tasks["end"] - tasks["begin"]).astype(np.int64)
    assert p.s["n_dense"] + p.s["n_pset"] > 0 and p.s["n_cand"] > 0
    dense = sched[: p.s["n_dense"] + p.s["n_pset"]][: max(p.s["n_dense"], 1) if p.s["n_dense"] else p.s["n_pset"]]
    assert span[dense[0]] >= span[dense[-1]]
    p.close()


def test_candidate_tile_tasks_are_queued_per_xcd_by_the_row_they_probe(world):
    """k_and's section of the schedule is cut into one queue per XCD: every task in exactly one queue.  In a batch of two-term conjunctions with long leads
    (`cand_xcd`, the default) a queue is ordered by the plane row its tasks probe first: a row sits in ONE queue — a row that outweighs a 16th of the
    section in a few — its tasks next to each other, and the queues carry about the same number of tiles; without it the cost order is dealt round the
    queues.  The set of tasks is the same either way, and so is the plan whatever the number of planning threads."""
    D, V, segs, hix = world
    parts, _ = W.build_parts("cfg2", D, V, 10, 42, 4000)
    seen = {}
    for xcd, threads in ((1, 4), (1, 1), (0, 4)):
        p = HP.HostPlan(hix[parts[0].codec], parts[0].programs, parts[0].flags, 0, threads=threads, options={"cand_xcd": xcd, "plane_div": 64})
        s, tasks, plan, qpl = p.s, p.tasks, p.plan, p.qplane
        c0, nc = s["n_dense"] + s["n_pset"] + s["n_probe"], s["n_cand"]
        cq = p.cand_q.astype(np.int64)
        assert nc >= 64 and cq[0] == 0 and cq[8] == nc and np.all(np.diff(cq) >= 0)
        sec = p.sched[c0 : c0 + nc].copy()
        assert np.all(tasks["kind"][sec] == HP.TASK_CAND)
        if (xcd, threads) == (1, 1):
            assert np.array_equal(sec, seen[1, 4])  # (the queues do not depend on how the batch was cut for planning)
        seen[xcd, threads] = sec
        tiles = (tasks["end"] - tasks["begin"]).astype(np.int64)[sec] + 1
        loads = np.array([tiles[cq[x] : cq[x + 1]].sum() for x in range(8)])
        assert loads.max() <= 1.25 * loads.mean() + 8, loads
        if not xcd:
            assert np.all(np.diff(cq) >= nc // 8) and np.all(np.diff(cq) <= nc // 8 + 1)
            p.close()
            continue
        rows = np.full(nc, -1, dtype=np.int64)
        for i, ti in enumerate(sec):
            q = plan[tasks["slot"][ti]]
            for k in range(1, int(q["nterms"])):
                r = int(qpl[int(q["term_base"]) + k]) if len(qpl) else 0xFFFFFFFF
                if r != 0xFFFFFFFF:
                    rows[i] = r
                    break
        assert (rows >= 0).sum() > nc // 4  # (the world's head terms have planes at this plane_div)
        queue_of = np.searchsorted(cq, np.arange(nc), side="right") - 1
        total = tiles[rows >= 0].sum()
        contiguous = 0
        for r in np.unique(rows[rows >= 0]):
            at = np.flatnonzero((rows == r) & (tiles <= 4))  # (the long tasks go first, whatever they probe)
            if not len(at):
                continue
            qs = np.unique(queue_of[at])
            assert len(qs) <= max(1, -(-16 * tiles[rows == r].sum() // total)), (r, qs)
            contiguous += all(a[-1] - a[0] + 1 == len(a) for a in (at[queue_of[at] == x] for x in qs))
        assert contiguous >= min(8 * 30, len(np.unique(rows[rows >= 0]))) // 2  # (30 places per queue: the lightest rows of a crowded queue share one)
        p.close()
    assert np.array_equal(np.sort(seen[0, 4]), np.sort(seen[1, 4]))


def test_unsupported_shapes_and_malformed_programs(world):
    D, V, segs, hix = world
    import oracle_lib as O

    big = " OR ".join(f"(t{2 * i} t{2 * i + 1})" for i in range(40))  # more than 64 nodes: still left out
    texts = ["t0 t1", 't0 OR "t1 t2"', big, "t0 OR (t1 t2) OR (t3 t4) OR (t5 t6) OR (t7 t8)", '[t0, "t1 t2", "t2 t3 t4"]']
    progs = [O.parse_query(t, some_min=2) for t in texts] * 700  # (enough queries for several fragments)
    for flags, topk in ((T.FLAG_DOCUMENTS_ONLY, 0), (T.FLAG_ACCUMULATED_SCORE, 10), (T.FLAG_MATCHED_TERMS, 0)):
        p = HP.HostPlan(hix[1], progs, flags, topk, threads=4)
        assert p.qstatus.tolist() == [0, 0, -3, 0, 0] * 700 and p.s["unsupported_queries"] == 700
        assert p.s["tree_queries"] == 2100 and p.s["n_tree_hidden"] == 2100 and p.s["n_plan"] == 2800 + 2100  # (three phrase leaves per five queries)
        check_plan(p, len(progs))
        p1 = HP.HostPlan(hix[1], progs, flags, topk, threads=1)
        assert np.array_equal(p1.block, p.block)
        p1.close()
        p.close()
    bad = progs[:2000] + [np.array([T.tok(T.OP_AND, 2)], dtype=np.uint32)] + progs[:100]
    with pytest.raises(T.TrinityError, match="malformed"):
        HP.HostPlan(hix[1], bad, T.FLAG_DOCUMENTS_ONLY, threads=4)


def test_plane_budget_caps_the_eligible_terms(world):
    D, V, segs, hix = world
    parts, _ = W.build_parts("cfg3", D, V, 10, 42, 2000)
    pt = parts[0]
    rows = []
    for budget in (1 << 40, 2_000_000, 300_000):
        p = HP.HostPlan(hix[pt.codec], pt.programs, pt.flags, pt.topk, threads=2, options={"plane_div": 1 << 30, "dense_min_postings": 0, "plane_max_bytes": budget})
        check_plan(p, len(pt.programs))
        assert p.s["n_plane_terms"] * 3 * p.s["plw"] * 4 <= max(budget, 3 * p.s["plw"] * 4)
        assert p.s["n_qplane"] == 0 or int(p.qplane[p.qplane != 0xFFFFFFFF].max()) < max(1, budget // (3 * p.s["plw"] * 4))  # rows stay inside the budget's rows
        rows.append(p.s["n_plane_terms"])
        p.close()
    assert rows[0] > rows[1] > rows[2] >= 1


def test_plane_set_tasks_name_a_row_for_every_term(world):
    D, V, segs, hix = world
    parts, _ = W.build_parts("cfg2", D, V, 10, 42, 4000)
    p = HP.HostPlan(hix[parts[0].codec], parts[0].programs, parts[0].flags, 0, threads=4, options={"dense_min_postings": 0, "plane_div": 64})
    check_plan(p, 4000)
    plan, tasks = p.plan, p.tasks
    assert p.s["n_pset"] > 0 and p.s["n_dense"] > 0  # (both kinds of bitmap-window queries in one batch: some partner lists are too short for a plane)
    kind_q = tasks["kind"][plan["first_task"]]
    for sidx in np.nonzero(kind_q == HP.TASK_PSET)[0][:200]:
        q = plan[sidx]
        assert np.all(p.qplane[q["term_base"] : q["term_base"] + q["nterms"]] != 0xFFFFFFFF)
    for sidx in np.nonzero(kind_q == HP.TASK_DENSE)[0][:200]:
        q = plan[sidx]
        assert np.any(p.qplane[q["term_base"] : q["term_base"] + q["nterms"]] == 0xFFFFFFFF)
    p.close()


def test_recycled_fragments_plan_the_same(world):
    """tri_dev keeps the planner's per-fragment arrays from plan to plan (planner.hpp: Frag::recycle, FragCache).  A plan made on buffers that earlier plans
    of OTHER shapes left behind — another workload, the other codec, another thread count — is byte for byte the plan made on fresh ones, with the same
    counters."""
    D, V, segs, hix = world
    order = [("cfg3", 2), ("cfg2", 1), ("cfg5", 1), ("cfg5", 2), ("cfg4", 1), ("cfg2", 1), ("cfg3", 2)]
    for wl, codec in order:
        parts, _ = W.build_parts(wl, D, V, 10, 42, 3000 if wl != "cfg4" else 700)
        for pt in parts:
            if pt.codec != codec:
                continue
            for threads in (8, 1):
                fresh = HP.HostPlan(hix[codec], pt.programs, pt.flags, pt.topk, threads=threads)
                reused = HP.HostPlan(hix[codec], pt.programs, pt.flags, pt.topk, threads=threads, options={"frag_cache": 1})
                assert bytes(fresh.block) == bytes(reused.block), (wl, codec, threads)
                assert dict(fresh.s) == dict(reused.s), (wl, codec, threads)
                fresh.close()
                reused.close()


def test_random_token_streams_are_planned_or_refused(world):
    """The planner is the C-ABI's front door: any sequence of tokens — operands missing, counts of zero, term ids past the dictionary, thresholds above the
    operand count, phrases of operators — is either lowered into a plan that satisfies every invariant or refused with TRI_ERR_INVALID; it never crashes
    (this test also runs under ASan + UBSan: tests/test_oracle_sanitized.py).  Well-formed random trees among them must come out as plans."""
    D, V, segs, hix = world
    rng = np.random.default_rng(99)

    def random_program(wellformed):
        if wellformed:  # a random postfix tree over a few terms
            st, out = 0, []
            for _ in range(int(rng.integers(1, 14))):
                if st >= 2 and rng.random() < 0.45:
                    op = int(rng.choice([T.OP_AND, T.OP_OR, T.OP_NOT, T.OP_OPT, T.OP_SOME]))
                    n = 2 if op in (T.OP_NOT, T.OP_OPT) else int(rng.integers(2, st + 1))
                    out.append(T.tok(op, n | ((int(rng.integers(1, n + 1)) << 16) if op == T.OP_SOME else 0)))
                    st -= n - 1
                else:
                    out.append(T.tok(T.OP_TERM, int(rng.integers(0, 40))))
                    st += 1
            if st > 1:
                out.append(T.tok(T.OP_AND, st))
            return np.array(out, dtype=np.uint32)
        n = int(rng.integers(1, 12))
        ops = rng.integers(0, 8, size=n)  # (7: not an operator at all)
        args = np.where(rng.random(n) < 0.7, rng.integers(0, 6, size=n), rng.integers(0, 1 << 28, size=n))
        return ((ops.astype(np.uint32) << 28) | args.astype(np.uint32)).astype(np.uint32)

    planned = refused = 0
    for flags, topk in ((T.FLAG_DOCUMENTS_ONLY, 0), (T.FLAG_ACCUMULATED_SCORE, 10), (T.FLAG_MATCHED_TERMS, 0)):
        for round_ in range(60):
            wellformed = round_ % 3 == 0
            progs = [random_program(wellformed) for _ in range(int(rng.integers(1, 40)))]
            try:
                p = HP.HostPlan(hix[1 + (round_ & 1)], progs, flags, topk, threads=1 + 3 * (round_ & 1))
            except T.TrinityError as e:
                assert not wellformed, (progs, str(e))
                refused += 1
                continue
            if p.s["n_plan"]:
                check_plan(p, len(progs))
            p.close()
            planned += 1
    assert planned >= 60 and refused >= 30


@pytest.mark.parametrize("codec", [T.engine.CODEC_GOOGLE, T.engine.CODEC_LUCENE])
def test_corrupted_segments_are_walked_or_refused(codec):
    """The upload walk (csrc/index_host.hpp: build_host_index — what tri_index_upload runs over the caller's bytes before anything reaches the device): a
    segment with flipped bytes, a truncated tail, or term-table entries that point elsewhere is either walked to a directory or refused with a message; it
    never reads outside the buffers it was given (the same test runs under ASan + UBSan)."""
    seg = T.Segment(6000, 300, 8, 5, codec=codec)
    index0, terms0 = np.array(seg.index, dtype=np.uint8), np.array(seg.terms, dtype=np.uint32).reshape(-1, 3)
    hits0 = np.array(seg.hits, dtype=np.uint8) if codec == T.engine.CODEC_LUCENE else None
    HP.HostIndex(index0, terms0, seg.docs_cnt, codec=codec, hits=hits0).close()  # (the untouched segment is accepted)
    rng = np.random.default_rng(17 + codec)
    walked = refused = 0
    for trial in range(250):
        index, terms = index0.copy(), terms0.copy()
        hits = None if hits0 is None else hits0.copy()
        kind = trial % 5
        if kind == 0:  # random bytes overwritten
            at = rng.integers(0, index.size, size=int(rng.integers(1, 9)))
            index[at] = rng.integers(0, 256, size=at.size)
        elif kind == 1:  # the tail cut off
            index = index[: int(rng.integers(0, index.size))].copy()
        elif kind == 2:  # a term's chunk moved / resized / its document count changed
            t = int(rng.integers(0, terms.shape[0]))
            terms[t, int(rng.integers(0, 3))] = int(rng.integers(0, 1 << 32, dtype=np.uint64))
        elif kind == 3:  # runs of 0xff / 0x00 (long varints, zero lengths)
            a = int(rng.integers(0, index.size - 64))
            index[a : a + int(rng.integers(1, 64))] = 0xFF if trial & 8 else 0
        elif hits is not None and hits.size:  # the hits file damaged
            at = rng.integers(0, hits.size, size=int(rng.integers(1, 9)))
            hits[at] = rng.integers(0, 256, size=at.size)
        else:
            terms[:, 1] = np.roll(terms[:, 1], 1)
        try:
            HP.HostIndex(index, terms, seg.docs_cnt, codec=codec, hits=hits).close()
            walked += 1
        except T.TrinityError as e:
            assert str(e)
            refused += 1
    assert refused >= 60 and walked + refused == 250


def test_upload_walk_takes_the_reference_written_segment_and_refuses_wandering_deltas():
    """The walk reads every interior delta: the reference-written edge segment (payloads, frequency 0, a 70000-hit document, positions up to MaxPosition - 1)
    is accepted as before; the same bytes with ONE interior delta of a full block enlarged — the block's documents now run past the block's last
    document, which the kernels' bitmaps rely on — are refused."""
    from test_oracle import load_edge

    g, index, terms = load_edge()
    index = np.frombuffer(bytes(index), dtype=np.uint8).copy()
    terms = np.array(terms, dtype=np.uint32).reshape(-1, 3)
    HP.HostIndex(index, terms, g["docsCnt"]).close()
    # a plain synthetic segment: the first block of the longest list is [skiplist count u16][delta varint][length varint][n = 32][31 one-byte deltas ...]
    seg = T.Segment(20000, 500, 8, 3)
    idx, tt = np.array(seg.index, dtype=np.uint8), np.array(seg.terms, dtype=np.uint32).reshape(-1, 3)
    t = int(np.argmax(tt[:, 0]))
    base = int(tt[t, 1]) + 2
    p = base
    for _ in range(2):  # the block header's two prefix varints
        b0 = int(idx[p])
        p += 1 if b0 < 0x80 else 2 if b0 < 0xC0 else 3 if b0 < 0xE0 else 4 if b0 < 0xF0 else 5
    assert idx[p] == 32 and np.all(idx[p + 1 : p + 32] < 0x80)  # (a head term: a full block of one-byte deltas)
    bad = idx.copy()
    bad[p + 5] = 0x7F  # one document pushed far beyond the block's last
    with pytest.raises(T.TrinityError, match="run past its last document"):
        HP.HostIndex(bad, tt, seg.docs_cnt)
    bad = idx.copy()
    bad[p + 5] = 0  # ... or repeated
    with pytest.raises(T.TrinityError, match="repeats inside a block"):
        HP.HostIndex(bad, tt, seg.docs_cnt)
    HP.HostIndex(idx, tt, seg.docs_cnt).close()
