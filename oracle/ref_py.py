"""Driver of oracle/_ref/ref_query: the reference's OWN indexing and shard query code (oracle/ref_query.c).
TEST INFRASTRUCTURE ONLY.  The binary exists where `make ref-query` ran (the build container); it is self-contained and
travels with the tree, /root/reference does not.

run(index, pqs) feeds the documents of a Python-built Index (marlin_amd.index.Index.add_documents keeps them per shard,
in docid order) to the reference's index_string / index_number / ... leaf calls in the order parse_index_document makes
them, then every parsed query as the fields index_json_query fills, and returns what the reference's execute_squery left
in struct squery_result, per query and shard."""
import ctypes as C
import os
import struct
import subprocess
import tempfile

from marlin_amd import ffi
from marlin_amd.index import F_BOOLEAN, F_NUMBER, F_NUMLIST, F_STRING, F_STRLIST, _value_type

_HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(_HERE, "_ref", "ref_query")
EXE_GPU = os.path.join(_HERE, "_ref", "ref_query_gpu")

_FT = {ffi.MT_STRING: 1, ffi.MT_NUMBER: 3, ffi.MT_BOOLEAN: 5, ffi.MT_OTHER: 0}      # mapping.h F_TYPE


def available(gpu=False):
    return os.path.exists(EXE_GPU if gpu else EXE)


def _hx(s):
    return s.encode("utf-8").hex() or "00"[:0]


def index_commands(ix, workdir, skip=None):
    """skip: documents per shard that are already indexed (a second bulk write appends the rest)"""
    out = [] if skip else ["I %s %d %d %d %d %d" % (workdir, ix.num_shards, len(ix.strings), len(ix.numbers), len(ix.bools), len(ix.facets))]
    for s in range(ix.num_shards):
        out.append("W %d" % s)
        for docid, d in enumerate(ix.docs[s], 1):
            if skip and docid <= skip[s]:
                continue
            out.append("D %d" % docid)
            for k, v in d.items():                           # parse_index_document, sindex.c:792-893
                f = ix.fields.get(k)
                if f is None or not (f.is_indexed or f.is_facet) or _value_type(v) != f.type:
                    continue
                if f.type in (F_STRING, F_STRLIST):
                    for e in ([v] if f.type == F_STRING else v):
                        if not isinstance(e, str):
                            continue
                        if f.is_facet:
                            out.append("f %d %s" % (f.f_priority, _hx(e)))
                        if f.is_indexed and e:
                            out.append("s %d %s" % (f.i_priority, _hx(e)))
                elif f.type in (F_NUMBER, F_NUMLIST):
                    for e in ([v] if f.type == F_NUMBER else v):
                        if isinstance(e, bool) or not isinstance(e, (int, float)):
                            continue
                        if f.is_indexed:                     # (od->num_data is only sized for indexed numbers)
                            out.append("n %d 1 %r" % (f.i_priority, float(e)))
                        if f.is_facet:
                            out.append("g %d %r" % (f.f_priority, float(e)))
                elif f.type == F_BOOLEAN and f.is_indexed:
                    out.append("b %d %d" % (f.i_priority, 1 if v else 0))
            out.append("E")
        out.append("C")
    return out


def _filter_lines(pq, node, depth, out):
    c = node.contents
    sv = pq.filter_strs.get(C.addressof(c))
    out.append("F %d %d %d %d %d %d %r %r %d %d %s" % (depth, c.type, _FT[c.field_type], c.i_priority, c.f_priority, c.is_facet,
                                                       c.numval, c.numval2, c.numcmp1, c.numcmp2, _hx(sv) if sv else "-"))
    for i in range(c.num_children):
        _filter_lines(pq, c.children[i], depth + 1, out)


def query_commands(ix, pq):
    q = pq.q.contents
    cfg = q.cfg
    out = ["Q"]
    text = pq.jq.get("q")
    if isinstance(text, str):
        out.append("t %s" % _hx(text))
    out.append("c %d %d %d %d %d %d %d %d %d %d %d" % (cfg.hits_per_page, q.page_num, cfg.max_hits, cfg.max_facet_results, cfg.prefix, cfg.typos,
                                                      cfg.rank_by, cfg.rank_sort, cfg.rank_asc, cfg.full_scan, cfg.full_scan_threshold))
    out.append("r %d %s" % (cfg.num_rules, " ".join(str(cfg.rank_algo[i]) for i in range(cfg.num_rules))))
    out.append("e %x" % sum(1 << f for f in range(len(ix.facets)) if cfg.facet_enabled[f]))
    if q.filter:
        _filter_lines(pq, q.filter, 0, out)
    for a in range(q.num_aggs):
        out.append("A %d %d" % (q.aggs[a].type, q.aggs[a].field))
    out.append("R")
    return out


def _f64(hexbits):
    return struct.unpack("<d", struct.pack("<Q", int(hexbits, 16)))[0]


def parse_output(text, num_shards, nq):
    """-> [query][shard] = {num_hits, rank_count, fast, ranks [(docid, typos, proximity, field, position, exact, compare)],
    facets {f: [(facet_id, count)]}, aggs [...]}"""
    res, cur = [], None
    for ln in text.splitlines():
        t = ln.split()
        if not t:
            continue
        if t[0] == "shard":
            cur = {"num_hits": int(t[3]), "rank_count": int(t[5]), "fast": int(t[7]), "ranks": [], "facets": {}, "aggs": []}
            fcur = None
        elif t[0] == "rank":
            cur["ranks"].append((int(t[1]), int(t[2]), int(t[3]), int(t[4]), int(t[5]), int(t[6]), _f64(t[7])))
        elif t[0] == "facet":
            fcur = cur["facets"].setdefault(int(t[1]), [])
        elif t[0] == "fc":
            fcur.append((int(t[1]), int(t[2])))
        elif t[0] == "agg":
            cur["aggs"].append({"type": int(t[2]), "v0": _f64(t[3]), "count": int(t[4]), "min": _f64(t[5]), "max": _f64(t[6]), "card": int(t[7])})
        elif t[0] == "end":
            res.append(cur)
    assert len(res) == num_shards * nq, (len(res), num_shards, nq, text[-2000:])
    return [res[i * num_shards:(i + 1) * num_shards] for i in range(nq)]


def run(ix, pqs, gpu=False, env=None, timeout=600):
    with tempfile.TemporaryDirectory() as tmp:
        cmds = index_commands(ix, os.path.join(tmp, "ix"))
        if gpu:
            cmds.append("X")
        for pq in pqs:
            cmds += query_commands(ix, pq)
        p = subprocess.run([EXE_GPU if gpu else EXE], input="\n".join(cmds) + "\n", text=True, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=timeout, env=env)
    if p.returncode != 0:
        raise RuntimeError("ref_query exited with %d\n%s\n%s" % (p.returncode, p.stdout[-2000:], p.stderr[-4000:]))
    return parse_output(p.stdout, ix.num_shards, len(pqs))


def run_two_writes(ix_a, ix_ab, pqs_a, pqs_ab, gpu=False, timeout=600, no_delta=False):
    """a second bulk write after the first queries: ix_ab holds ix_a's documents and more, in the same shards and order.
    gpu: the mirror is attached after the first write and REFRESHED (mgpu_shim_refresh_shard) after the second -- by a
    delta (the call returns 0), or with no_delta (MGPU_SHIM_NO_DELTA=1) by a rebuild next to the live mirror (2).
    Returns (results of pqs_a on the first state, results of pqs_ab on the second)."""
    for s in range(ix_a.num_shards):
        assert ix_ab.docs[s][:len(ix_a.docs[s])] == ix_a.docs[s]
    with tempfile.TemporaryDirectory() as tmp:
        cmds = index_commands(ix_a, os.path.join(tmp, "ix"))
        if gpu:
            cmds.append("X")
        for pq in pqs_a:
            cmds += query_commands(ix_a, pq)
        cmds += index_commands(ix_ab, None, skip=[len(d) for d in ix_a.docs])
        if gpu:
            cmds.append("Y")
        for pq in pqs_ab:
            cmds += query_commands(ix_ab, pq)
        env = dict(os.environ)
        if no_delta:
            env["MGPU_SHIM_NO_DELTA"] = "1"
        p = subprocess.run([EXE_GPU if gpu else EXE], input="\n".join(cmds) + "\n", text=True, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=timeout, env=env)
    if p.returncode != 0:
        raise RuntimeError("ref_query exited with %d\n%s\n%s" % (p.returncode, p.stdout[-2000:], p.stderr[-4000:]))
    allres = parse_output(p.stdout, ix_a.num_shards, len(pqs_a) + len(pqs_ab))
    refreshed = [ln.split() for ln in p.stdout.splitlines() if ln.startswith("refresh ")]
    if gpu:
        assert len(refreshed) == ix_a.num_shards and all(t[2] == ("2" if no_delta else "0") for t in refreshed), refreshed
    return allres[:len(pqs_a)], allres[len(pqs_a):]
