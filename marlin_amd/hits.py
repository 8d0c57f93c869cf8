"""Row f-4: hit materialisation (query.c:385-401) -- the stored document of every hit on the page, projected by
getFields (json_handle_limit_field, query.c:155-187) and highlighted (hit_query_processing, query.c:189-310) with
m_highlight (marlin_amd/host/mhighlight.c = highlight.c restated).

The reference keeps documents as zlib-compressed JSON in LMDB (sdata.c:183-201) and walks them with jansson; here the
index keeps the parsed documents (Index.docs) and the walk over the JSON tree is Python -- plumbing around the one
compute piece, the string matcher, which is C.  Settings follow parse_query_settings (index.c:573-617): getFields and
highlightFields are arrays of dotted paths, ["*"] (the default, index.c:1376) highlights every string, null none;
highlightSource puts the marks into the hit itself instead of a copy under "_highlight".
"""
import copy
import ctypes as C

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
                h = self.text(value)
                if h is not None:
                    j[key] = h
            elif isinstance(value, dict):
                self.everything(value)
            elif isinstance(value, list):
                for i, av in enumerate(value):
                    if isinstance(av, str):
                        h = self.text(av)
                        if h is not None:
                            value[i] = h
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
                h = self.text(value)
                if h is not None:
                    j[key] = h
            elif isinstance(value, dict):
                if child:
                    self.some(value, child)
            elif isinstance(value, list):
                for i, av in enumerate(value):
                    if isinstance(av, str):
                        h = self.text(av)
                        if h is not None:
                            value[i] = h
                    elif isinstance(av, dict) and child:
                        self.some(av, child)
        return j


def materialize(lib, pq, doc):
    """hit_query_processing (query.c:290-310) for one stored document -> the hit object of the response"""
    get_fields, hl, in_source = pq.hit_cfg
    hit = copy.deepcopy(doc)
    if not get_fields and hl is None:
        return hit
    if get_fields:
        hit = limit_fields(hit, get_fields)
    if hl is None:          # (the reference dereferences the NULL list here when getFields is set: no highlighting)
        return hit
    h = Highlighter(lib, pq.q)
    if in_source:
        return h.everything(hit) if hl == ALL else h.some(hit, hl)
    dup = copy.deepcopy(hit)
    dup = h.everything(dup) if hl == ALL else h.some(dup, hl)
    hit["_highlight"] = dup
    return hit
