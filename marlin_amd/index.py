"""Index / shard / query objects over the C host library.

Mirrors the reference's index-level behaviour needed to feed the shard-query path:
schema discovery and priorities (mapping.c:537-601), per-query settings overrides
(index.c:522-630), filter parsing (filter.c:58-348), aggregation parsing (aggs.c:56-153,
metric-aggs.c:75-103,191-399).  JSON lives in Python here because the reference does this in
jansson on the host; none of it is on the GPU path.
"""
import ctypes as C
import zlib

from . import ffi, hits

F_STRING, F_STRLIST, F_NUMBER, F_NUMLIST, F_BOOLEAN, F_NULL = "string", "strlist", "number", "numlist", "boolean", "null"


class QueryError(ValueError):
    """the reference answers HTTP 400 for these (index.c:1083-1127)"""


class Field:
    def __init__(self, name):
        self.name = name
        self.type = F_NULL
        self.is_indexed = False
        self.is_facet = False
        self.i_priority = 0
        self.f_priority = 0


def _value_type(v):
    if isinstance(v, bool):
        return F_BOOLEAN
    if isinstance(v, (int, float)):
        return F_NUMBER
    if isinstance(v, str):
        return F_STRING
    if isinstance(v, list) and v:
        t = _value_type(v[0])
        if t == F_STRING:
            return F_STRLIST
        if t == F_NUMBER:
            return F_NUMLIST
    return F_NULL


class Index:
    """settings: {"indexedFields": [...], "facetFields": [...], + default query settings}"""

    def __init__(self, settings, num_shards=1):
        self.lib = ffi.load()
        self.settings = dict(settings)
        self.num_shards = num_shards
        self.fields = {}
        self.strings, self.numbers, self.bools, self.facets = [], [], [], []
        self.shards = []          # mshard_t* (c_void_p)
        self.docs = []            # per shard: list of original docs, docid-1 indexed
        self.stored = []          # per shard: the docid2json values of the same documents (hits.pack_stored), docid-1 indexed
        self.mirrors = None
        self._ready = False

    # -- schema (mapping_apply_config, mapping.c:537-601)
    def _learn(self, docs):
        for d in docs:
            for k, v in d.items():
                t = _value_type(v)
                if t == F_NULL:
                    continue
                f = self.fields.setdefault(k, Field(k))
                if f.type == F_NULL:
                    f.type = t

    def _apply_config(self):
        for name in self.settings.get("indexedFields", []):
            f = self.fields.get(name)
            if f is None or f.type == F_NULL:
                raise ValueError("indexed field %r has no learnt type" % name)
            f.is_indexed = True
            if f.type in (F_STRING, F_STRLIST):
                f.i_priority = len(self.strings); self.strings.append(name)
            elif f.type in (F_NUMBER, F_NUMLIST):
                f.i_priority = len(self.numbers); self.numbers.append(name)
            elif f.type == F_BOOLEAN:
                f.i_priority = len(self.bools); self.bools.append(name)
        for name in self.settings.get("facetFields", []):
            f = self.fields.get(name)
            if f is None or f.type == F_NULL:
                raise ValueError("facet field %r has no learnt type" % name)
            f.is_facet = True
            f.f_priority = len(self.facets); self.facets.append(name)

    def field_schema(self, name):
        """get_field_schema: only fields of the index schema (indexed or facet)"""
        f = self.fields.get(name)
        if f is None or not (f.is_indexed or f.is_facet):
            return None
        return f

    def route(self, doc, ordinal):
        if self.num_shards == 1:
            return 0
        if "_id" in doc:      # stand-in for farmhash(_id) % num_shards (utils.c:59-62)
            return zlib.crc32(str(doc["_id"]).encode()) % self.num_shards
        return ordinal % self.num_shards

    def add_documents(self, docs):
        """builds the shards; call once with the whole corpus"""
        lib = self.lib
        self._learn(docs)
        self._apply_config()
        builders = [lib.msb_new(len(self.strings), len(self.numbers), len(self.bools), len(self.facets))
                    for _ in range(self.num_shards)]
        self.docs = [[] for _ in range(self.num_shards)]
        self.stored = [[] for _ in range(self.num_shards)]
        for n, d in enumerate(docs):
            s = self.route(d, n)
            b = builders[s]
            lib.msb_begin_doc(b)
            self.docs[s].append(d)
            self.stored[s].append(hits.pack_stored(d))
            for k, v in d.items():                       # parse_index_document, sindex.c:792-893
                f = self.fields.get(k)
                if f is None or not (f.is_indexed or f.is_facet) or _value_type(v) != f.type:
                    continue
                if f.type == F_STRING:
                    if f.is_facet:
                        lib.msb_add_facet_str(b, f.f_priority, v.encode())
                    if f.is_indexed:
                        lib.msb_add_string(b, f.i_priority, v.encode())
                elif f.type == F_STRLIST:
                    for e in v:
                        if not isinstance(e, str):
                            continue
                        if f.is_facet:
                            lib.msb_add_facet_str(b, f.f_priority, e.encode())
                        if f.is_indexed:
                            lib.msb_add_string(b, f.i_priority, e.encode())
                elif f.type == F_NUMBER:
                    if f.is_indexed:
                        lib.msb_add_number(b, f.i_priority, float(v), 1)
                    if f.is_facet:
                        lib.msb_add_facet_num(b, f.f_priority, float(v))
                elif f.type == F_NUMLIST:
                    for e in v:
                        if isinstance(e, bool) or not isinstance(e, (int, float)):
                            continue
                        if f.is_indexed:
                            lib.msb_add_number(b, f.i_priority, float(e), 1)
                        if f.is_facet:
                            lib.msb_add_facet_num(b, f.f_priority, float(e))
                elif f.type == F_BOOLEAN:
                    if f.is_indexed:
                        lib.msb_add_bool(b, f.i_priority, 1 if v else 0)
            lib.msb_end_doc(b)
        self.shards = [C.c_void_p(lib.msb_finalize(b)) for b in builders]
        self._ready = True
        return self

    @classmethod
    def from_shards(cls, settings, shards, strings, numbers, bools, facets):
        """wraps shards built in C (synthetic corpora); field lists give the priorities"""
        ix = cls(settings, num_shards=len(shards))
        for lst, typ, attr in ((strings, F_STRING, "strings"), (numbers, F_NUMBER, "numbers"), (bools, F_BOOLEAN, "bools")):
            for i, name in enumerate(lst):
                f = ix.fields.setdefault(name, Field(name))
                f.type = typ; f.is_indexed = True; f.i_priority = i
                getattr(ix, attr).append(name)
        for i, name in enumerate(facets):
            f = ix.fields.setdefault(name, Field(name))
            if f.type == F_NULL:
                f.type = F_STRING
            f.is_facet = True; f.f_priority = i
            ix.facets.append(name)
        ix.shards = [C.c_void_p(s) for s in shards]
        ix.docs = [None] * len(shards)
        ix._ready = True
        return ix

    def max_docid(self, shard):
        return C.cast(self.shards[shard], C.POINTER(ffi.ShardStruct)).contents.max_docid

    def facet_string(self, shard, facet_id):
        s = self.lib.mshard_lookup_facet(self.shards[shard], facet_id)
        return s.decode() if s is not None else None

    def close(self):
        if self.mirrors:
            for m in self.mirrors:
                self.lib.mmirror_destroy(m)
            self.mirrors = None
        for s in self.shards:
            self.lib.mshard_free(s)
        self.shards = []

    # -- queries ------------------------------------------------------------------------
    def query(self, jq):
        return ParsedQuery(self, jq)


class ParsedQuery:
    """index_json_query's parsing half (index.c:1072-1156): owns an m_query_t"""

    def __init__(self, index, jq):
        lib = index.lib
        self.index = index
        self.jq = jq
        self.q = lib.mq_new(len(index.facets))
        self.explain = bool(jq.get("explain", False))
        self.filter_strs = {}
        self.agg_names = []
        q = self.q.contents
        try:
            merged = dict(index.settings)
            merged.update(jq)
            self._settings(q, merged)
            try:
                self.hit_cfg = hits.hit_settings(merged)          # getFields / highlightFields / highlightSource
            except ValueError as e:
                raise QueryError(str(e))
            jf = jq.get("filter")
            if isinstance(jf, dict) and jf:          # index.c:1091-1093: anything but a non-empty object is no filter
                f = self._parse_filter(jf)
                q.filter = f
            ja = jq.get("aggs")
            if isinstance(ja, dict) and ja:          # index.c:1107-1109, same rule
                self._parse_aggs(ja)
            page = jq.get("page")
            if isinstance(page, int) and not isinstance(page, bool):
                if page <= 0 or page > 0xFFFF:
                    raise QueryError("Page number requested over limit")
                q.page_num = page
            if q.page_num <= 0 or q.page_num * q.cfg.hits_per_page > q.cfg.max_hits:
                raise QueryError("Page number requested over limit")
            lib.mq_build_rank_rule(self.q)
            text = jq.get("q")
            if isinstance(text, str):
                if lib.mq_set_text(self.q, text.encode()) != 0:
                    raise QueryError("query has more than %d words" % ffi.M_MAX_QWORDS)
        except Exception:
            lib.mq_free(self.q)
            self.q = None
            raise

    def close(self):
        if self.q is not None:
            self.index.lib.mq_free(self.q)
            self.q = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # parse_query_settings (index.c:522-630)
    def _settings(self, q, j):
        cfg = q.cfg
        if "hitsPerPage" in j: cfg.hits_per_page = int(j["hitsPerPage"])
        if "maxHits" in j: cfg.max_hits = int(j["maxHits"])
        if "maxFacetResults" in j: cfg.max_facet_results = int(j["maxFacetResults"]) & 0xFFFF
        if "rankAlgorithm" in j:
            ra = j["rankAlgorithm"]
            if not isinstance(ra, list):
                raise QueryError("Rank algorithm needs to be an array")
            for i, s in enumerate(ra):
                if not isinstance(s, str):
                    raise QueryError("Rank alorithms must be string")
                if s not in ffi.RULE_NAMES:
                    raise QueryError("Unknown rank rule specified")
                cfg.rank_algo[i] = ffi.RULE_NAMES[s]
            cfg.num_rules = len(ra)
        for key, sort in (("rankBy", False), ("sortBy", True)):
            if key in j:
                jo = j[key]
                cfg.rank_sort = sort
                for k, v in jo.items():
                    if v == "asc": cfg.rank_asc = True
                    elif v == "desc": cfg.rank_asc = False
                    else: raise QueryError("Invalid sort / rank direction specified")
                    f = self.index.field_schema(k)
                    if f is None:
                        raise QueryError("The field to sort / rank by is not indexed")
                    if f.type not in (F_NUMBER, F_NUMLIST):
                        raise QueryError("The field to sort / rank by is not a number")
                    cfg.rank_by = f.i_priority
        if "fullScan" in j: cfg.full_scan = j["fullScan"] is True
        if "fullScanThreshold" in j: cfg.full_scan_threshold = int(j["fullScanThreshold"])
        if "typos" in j: cfg.typos = ffi.MTYPO_OK if j["typos"] else ffi.MTYPO_NONE
        if "prefix" in j: cfg.prefix = {"last": ffi.MPREFIX_LAST, "all": ffi.MPREFIX_ALL, "none": ffi.MPREFIX_NONE}[j["prefix"]]
        if cfg.hits_per_page > cfg.max_hits:
            raise QueryError("Hits per page cannot be more than maximum hits")
        if cfg.max_hits > 1000:
            raise QueryError("Maximum hits cannot be more than 1000")
        if cfg.full_scan_threshold < 5000:
            raise QueryError("Full scan threshold cannot be less than 5000")

    # ---- filters (filter.c) ----
    def _new(self, ftype):
        return self.index.lib.mf_new(ftype)

    def _err(self, msg):
        raise QueryError(msg)

    def _validate(self, s, v):                     # validate_schema_json, filter.c:58-80
        if isinstance(v, str):
            if s.type not in (F_STRING, F_STRLIST): self._err("String value for non-string field " + s.name)
            if not s.is_facet: self._err("String filter for non-faceted string field " + s.name)
        elif isinstance(v, bool):
            if s.type != F_BOOLEAN: self._err("Boolean value for non-boolean field " + s.name)
        elif isinstance(v, (int, float)):
            if s.type not in (F_NUMBER, F_NUMLIST): self._err("Number value for non-numeric field " + s.name)

    def _leaf(self, s, ftype, v):
        f = self._new(ftype)
        c = f.contents
        c.i_priority = s.i_priority; c.f_priority = s.f_priority; c.is_facet = 1 if s.is_facet else 0
        c.field_type = ffi.MT_OTHER
        if isinstance(v, str):
            c.field_type = ffi.MT_STRING
            b = v.encode()
            c.facet_id = self.index.lib.m_facet_id(b, len(b))
            self.filter_strs[C.addressof(c)] = v          # filter.strval, for callers that want the string itself
        elif isinstance(v, bool):
            c.field_type = ffi.MT_BOOLEAN; c.numval = 1.0 if v else 0.0
        elif isinstance(v, (int, float)):
            c.field_type = ffi.MT_NUMBER; c.numval = float(v)
        return f

    _OPS = {"$eq": ffi.MF_EQ, "$ne": ffi.MF_NE, "$and": ffi.MF_AND, "$or": ffi.MF_OR, "$in": ffi.MF_OR,
            "$nin": ffi.MF_NIN, "$gt": ffi.MF_GT, "$gte": ffi.MF_GTE, "$lt": ffi.MF_LT, "$lte": ffi.MF_LTE}

    def _schema_array(self, s, arr, ftype):        # parse_schema_array, filter.c:82-104
        f = self._new(ftype)
        f.contents.field_type = ffi.MT_OTHER
        for v in arr:
            self.index.lib.mf_add_child(f, self._schema_json(s, v))
        return f

    def _schema_key(self, s, key, v):              # parse_schema_key_json, filter.c:106-166
        self._validate(s, v)
        ftype = self._OPS.get(key)
        if ftype is None:
            self._err("Invalid operator %s specified for field %s" % (key, s.name))
        if ftype in (ffi.MF_GT, ffi.MF_GTE, ffi.MF_LT, ffi.MF_LTE) and s.type not in (F_NUMBER, F_NUMLIST):
            self._err("Invalid operator %s specified for field %s" % (key, s.name))
        if ftype in (ffi.MF_OR, ffi.MF_NIN, ffi.MF_AND):
            if not isinstance(v, list):
                self._err("Array expected for operator %s field %s" % (key, s.name))
            return self._schema_array(s, v, ftype)
        if isinstance(v, list): self._err("Array not expected for operator %s field %s" % (key, s.name))
        if v is None: self._err("NULL not expected for operator %s field %s" % (key, s.name))
        return self._leaf(s, ftype, v)

    def _schema_object(self, s, obj):              # parse_schema_object, filter.c:173-225
        if len(obj) == 1:
            (k, v), = obj.items()
            return self._schema_key(s, k, v)
        f = self._new(ffi.MF_AND)
        c = f.contents
        c.field_type = ffi.MT_NUMBER if s.type == F_NUMBER else ffi.MT_OTHER
        kids = []
        for k, v in obj.items():
            ch = self._schema_key(s, k, v)
            cc = ch.contents
            if cc.type in (ffi.MF_GT, ffi.MF_GTE): c.numcmp1 = cc.type; c.numval = cc.numval
            if cc.type in (ffi.MF_LT, ffi.MF_LTE): c.numcmp2 = cc.type; c.numval2 = cc.numval
            kids.append(ch)
        if s.type in (F_NUMBER, F_NUMLIST) and len(kids) == 2:
            if kids[0].contents.type == ffi.MF_AND:
                self._err("Invalid and operation for numeric filter")
            c.type = ffi.MF_RANGE                   # range fusion, filter.c:210-221
            c.i_priority = s.i_priority
            for ch in kids:
                self.index.lib.mf_free(ch)
        else:
            for ch in kids:
                self.index.lib.mf_add_child(f, ch)
        return f

    def _schema_json(self, s, v):                  # parse_schema_json, filter.c:229-266
        self._validate(s, v)
        if isinstance(v, (str, bool, int, float)):
            return self._leaf(s, ffi.MF_EQ, v)
        if isinstance(v, dict):
            return self._schema_object(s, v)
        if isinstance(v, list):
            return self._schema_array(s, v, ffi.MF_AND)
        self._err("Json parse failure for field " + s.name)

    def _key_json(self, key, v):                   # parse_key_json, filter.c:269-314
        s = self.index.field_schema(key)
        if s is not None:
            return self._schema_json(s, v)
        ftype = self._OPS.get(key)
        if ftype not in (ffi.MF_OR, ffi.MF_NIN, ffi.MF_AND):
            self._err("Invalid field or operator " + key)
        if not isinstance(v, list):
            self._err("Array expected for operator " + key)
        f = self._new(ftype)
        for e in v:
            self.index.lib.mf_add_child(f, self._parse_filter(e))
        return f

    def _parse_filter(self, j):                    # parse_filter, filter.c:317-348
        if not isinstance(j, dict):
            self._err("Query filter json parsing error")
        if len(j) == 1:
            (k, v), = j.items()
            return self._key_json(k, v)
        f = self._new(ffi.MF_AND)
        for k, v in j.items():
            self.index.lib.mf_add_child(f, self._key_json(k, v))
        return f

    # ---- aggs (aggs.c:56-140, metric-aggs.c parse_*) ----
    def _parse_aggs(self, ja):
        self.agg_names = []
        kinds = {"max": ffi.MA_MAX, "min": ffi.MA_MIN, "avg": ffi.MA_AVG, "stats": ffi.MA_STATS,
                 "cardinality": ffi.MA_CARDINALITY}
        for name, spec in ja.items():
            found = None
            for k, v in spec.items():
                if k in kinds:
                    found = (kinds[k], v)
            if found is None:
                raise QueryError("Could not parse agg " + name)
            kind, v = found
            s = self.index.field_schema(v.get("field", "")) if isinstance(v, dict) else None
            if kind == ffi.MA_CARDINALITY:
                if s is None or not s.is_facet:
                    raise QueryError("Failed to parse card aggr %s, requires a facet value." % name)
                field = s.f_priority
            else:
                if s is None or not s.is_indexed or s.type not in (F_NUMBER, F_NUMLIST):
                    raise QueryError("Failed to parse aggr " + name)
                field = s.i_priority
            if self.index.lib.mq_add_agg(self.q, kind, field) != 0:
                raise QueryError("too many aggregations")
            self.agg_names.append(name)
