"""Row f-4: hit materialisation (query.c:385-401) -- the stored document of every hit on the page, projected by
getFields (json_handle_limit_field, query.c:155-187) and highlighted (hit_query_processing, query.c:189-310) with
m_highlight (marlin_amd/host/mhighlight.c = highlight.c restated).

The reference keeps documents as zlib-compressed JSON in LMDB (docid2json: [u32 length][compress() output],
sdata.c:105-122, read back by sdata_get_document_byid, sdata.c:183-201) and walks them with jansson; the index here keeps
its documents in exactly that value format (Index.stored, pack_stored / unpack_stored) and the walk over the JSON tree is
Python -- plumbing around the one compute piece, the string matcher.  One string at a time it runs on the host
(m_highlight); for the pages of a whole batch of queries PageBatch collects every string first and matches all their
tokens in one device launch (m_highlight_batch -> mgpu_highlight_match).  Settings follow parse_query_settings (index.c:573-617): getFields and
highlightFields are arrays of dotted paths, ["*"] (the default, index.c:1376) highlights every string, null none;
highlightSource puts the marks into the hit itself instead of a copy under "_highlight".
"""
import copy
import ctypes as C
import json
import struct
import zlib


def pack_stored(doc):
    """the docid2json value of a document (sdata_add_document, sdata.c:105-122): u32 length of the JSON text, then its
    zlib stream"""
    raw = json.dumps(doc, separators=(",", ":"), ensure_ascii=False).encode("utf-8")
    return struct.pack("<I", len(raw)) + zlib.compress(raw)


def unpack_stored(val):
    """sdata_get_document_byid (sdata.c:183-201) + json_loads"""
    (n,) = struct.unpack_from("<I", val, 0)
    raw = zlib.decompress(val[4:])
    if len(raw) != n:
        raise ValueError("stored document: length prefix %d, inflated %d" % (n, len(raw)))
    return json.loads(raw.decode("utf-8"))

ALL = "*"          # qcfg->highlight_fields with an empty name: every field


def parse_limit_fields(arr):
    """parse_limit_fields (index.c:459-519): ["a", "b.c", "b.d"] -> {"a": {}, "b": {"c": {}, "d": {}}}.
    A name is kept once; order of first appearance is the order the projection emits fields in."""
    root = {}
    for name in arr:
        if not isinstance(name, str):
            continue
        node = root
        for tok in [t for t in name.split(".") if t]:
            node = node.setdefault(tok, {})
    return root


def hit_settings(merged):
    """(get_fields, highlight_fields, highlight_source) from index settings overlaid with the query's"""
    gf = merged.get("getFields")
    if gf is not None and not isinstance(gf, list):
        raise ValueError("getFields needs to be a string array of fields to retrieve")
    get_fields = parse_limit_fields(gf) if gf is not None else None
    if "highlightFields" in merged:
        hf = merged["highlightFields"]
        if hf is None:
            hl = None
        elif not isinstance(hf, list):
            raise ValueError("highlightFields needs to be a string array of fields to highlight")
        elif len(hf) == 1 and hf[0] == "*":
            hl = ALL
        else:
            hl = parse_limit_fields(hf)
    else:
        hl = ALL
    return get_fields or None, hl, merged.get("highlightSource") is True


def limit_fields(doc, fields):
    """json_handle_limit_field (query.c:155-187)"""
    out = {}
    for name, child in fields.items():
        if not isinstance(doc, dict) or name not in doc:
            continue
        jo = copy.deepcopy(doc[name])
        if isinstance(jo, dict):
            out[name] = limit_fields(jo, child) if child else jo
        elif isinstance(jo, list):
            for i, av in enumerate(jo):
                if isinstance(av, dict) and child:
                    jo[i] = limit_fields(av, child)
            out[name] = jo
        else:
            out[name] = jo
    return out


class Highlighter:
    def __init__(self, lib, q):
        self.lib, self.q = lib, q

    def mark(self, container, key, s):
        """container[key] = the highlighted form of s when something matched"""
        h = self.text(s)
        if h is not None:
            container[key] = h

    def text(self, s):
        p = self.lib.m_highlight(s.encode("utf-8"), self.q, 0)
        if not p:
            return None
        try:
            return C.string_at(p).decode("utf-8", "replace")
        finally:
            self.lib.m_highlight_free(p)

    def everything(self, j):
        """highlight_json (query.c:190-218): every string, strings and objects inside arrays, objects recursively"""
        for key, value in list(j.items()):
            if isinstance(value, str):
                self.mark(j, key, value)
            elif isinstance(value, dict):
                self.everything(value)
            elif isinstance(value, list):
                for i, av in enumerate(value):
                    if isinstance(av, str):
                        self.mark(value, i, av)
                    elif isinstance(av, dict):
                        self.everything(av)
        return j

    def some(self, j, fields):
        """highlight_json_limit_field (query.c:241-275): only the named fields; objects only through named children"""
        for key, value in list(j.items()):
            if key not in fields:
                continue
            child = fields[key]
            if isinstance(value, str):
                self.mark(j, key, value)
            elif isinstance(value, dict):
                if child:
                    self.some(value, child)
            elif isinstance(value, list):
                for i, av in enumerate(value):
                    if isinstance(av, str):
                        self.mark(value, i, av)
                    elif isinstance(av, dict) and child:
                        self.some(av, child)
        return j


class PageBatch:
    """the strings of the pages of a batch of queries, highlighted together: materialize(..., batch=this) only records
    where each string sits; flush() cuts them all into tokens, matches every token against its query's terms in ONE
    device launch (m_highlight_batch -> mgpu_highlight_match) and puts the marked strings in place"""

    class _Recorder(Highlighter):
        def __init__(self, batch, pq):
            self.batch, self.pq = batch, pq

        def mark(self, container, key, s):
            self.batch.slots.append((container, key, s, self.pq))

    def __init__(self, lib, ctx):
        self.lib, self.ctx, self.slots = lib, ctx, []

    def recorder(self, pq):
        return PageBatch._Recorder(self, pq)

    def flush(self):
        from . import ffi
        n = len(self.slots)
        if not n:
            return 0
        strs = (C.c_char_p * n)(*[s.encode("utf-8") for _, _, s, _ in self.slots])
        qs = (C.POINTER(ffi.Query) * n)(*[pq.q for _, _, _, pq in self.slots])
        out = (C.c_void_p * n)()
        rc = self.lib.m_highlight_batch(self.ctx, strs, qs, None, n, out)
        if rc:
            raise RuntimeError("m_highlight_batch: %d %s" % (rc, self.lib.mgpu_last_error().decode()))
        marked = 0
        for (container, key, _, _), p in zip(self.slots, out):
            if p:
                container[key] = C.string_at(p).decode("utf-8", "replace")
                self.lib.m_highlight_free(p)
                marked += 1
        self.slots = []
        return marked


def materialize(lib, pq, doc, batch=None):
    """hit_query_processing (query.c:290-310) for one stored document -> the hit object of the response.  doc: the parsed
    document, or its stored value (bytes: unpack_stored).  batch: a PageBatch that highlights later, all strings at once."""
    get_fields, hl, in_source = pq.hit_cfg
    hit = unpack_stored(doc) if isinstance(doc, (bytes, bytearray)) else copy.deepcopy(doc)
    if not get_fields and hl is None:
        return hit
    if get_fields:
        hit = limit_fields(hit, get_fields)
    if hl is None:          # (the reference dereferences the NULL list here when getFields is set: no highlighting)
        return hit
    h = batch.recorder(pq) if batch is not None else Highlighter(lib, pq.q)
    if in_source:
        return h.everything(hit) if hl == ALL else h.some(hit, hl)
    dup = copy.deepcopy(hit)
    dup = h.everything(dup) if hl == ALL else h.some(dup, hl)
    hit["_highlight"] = dup
    return hit
