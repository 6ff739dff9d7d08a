/* CPython side of the drop-in class: the two loops of ClusterCRF.predict_probabilities that touch every Gene /
 * Protein / Domain object, written against the C API (the arithmetic is on the device; on a metagenome these loops
 * ARE the call).
 *
 *   pack_protein(contigs, attr_index)   gecco/crf/features.py:13-35 + [EXT] CRFsuite's attribute lookup:
 *       per gene the distinct domain names in domain order, mapped to attribute ids, unknown names dropped
 *       -> (item_ptr int64, attr_ptr int64, attr_id int32) as bytes objects (numpy.frombuffer on the Python side).
 *       Duck-typed: gene.protein.domains[*].name through getattr, any object model.
 *   annotate_all(genes, probs, w1, gene_cls, prot_cls, dom_cls)   features.py:74-96, model.py:364-375,
 *       crf/__init__.py:261-269: new Gene / Protein / Domain objects carrying the probability and the domains'
 *       cluster weights.  Plain dataclass models only (GECCO's, this package's): an object is its __dict__, so a copy
 *       is tp_alloc + PyDict_Copy + three stores.  Returns None when an object of another class turns up (the caller
 *       then goes through the objects' own with_* methods).
 *
 * Built by gecco_amd/build.py next to the HIP library (gcc, Python headers only; no numpy C API).  The Python
 * statements of both loops stay in packing.py / crf.py: they are what runs when this module cannot be built, and what
 * tests/test_host_logic.py compares it with. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static PyObject *s_protein, *s_domains, *s_name, *s_probability, *s_cluster_weight, *s_qualifiers, *s__probability, *s_copy;

/* growable raw buffers */
typedef struct {
    char *p;
    size_t len, cap;
} buf_t;
static int buf_push(buf_t *b, const void *src, size_t n)
{
    if (b->len + n > b->cap) {
        size_t cap = b->cap ? b->cap * 2 : 4096;
        while (cap < b->len + n) cap *= 2;
        char *q = (char *)realloc(b->p, cap);
        if (!q) {
            PyErr_NoMemory();
            return -1;
        }
        b->p = q;
        b->cap = cap;
    }
    memcpy(b->p + b->len, src, n);
    b->len += n;
    return 0;
}

/* getattr(o, name) for the plain objects these loops meet: when the type has nothing of that name that would take
 * precedence over the instance dictionary (no data descriptor: a property, a slot), the value is read from the instance
 * dictionary directly; anything else -- and a miss -- goes through PyObject_GetAttr.  New reference. */
static inline PyObject *attr_of(PyObject *o, PyObject *name)
{
    PyObject **dp = Py_TYPE(o)->tp_getattro == PyObject_GenericGetAttr ? _PyObject_GetDictPtr(o) : NULL; /* (no __getattribute__ of its own) */
    if (dp && *dp) {
        PyObject *descr = _PyType_Lookup(Py_TYPE(o), name); /* borrowed; the type's method cache answers */
        if (!descr || !Py_TYPE(descr)->tp_descr_set) {
            PyObject *v = PyDict_GetItemWithError(*dp, name); /* borrowed */
            if (v) {
                Py_INCREF(v);
                return v;
            }
            if (PyErr_Occurred()) return NULL;
        }
    }
    return PyObject_GetAttr(o, name);
}

/* a <= b (le) or a < b (lt) for two keys: machine integers and exact strings without a call, the rest by rich comparison.
 * Returns 1 / 0, or -1 with an exception set. */
static inline int key_cmp(PyObject *a, PyObject *b, int op)
{
    if (PyLong_CheckExact(a) && PyLong_CheckExact(b)) {
        int oa = 0, ob = 0;
        const long long x = PyLong_AsLongLongAndOverflow(a, &oa), y = PyLong_AsLongLongAndOverflow(b, &ob);
        if (!oa && !ob) return op == Py_LE ? x <= y : op == Py_LT ? x < y : x == y;
    }
    return PyObject_RichCompareBool(a, b, op);
}

/* the distinct, known domain names of one gene, in list order, as attribute ids (gecco/crf/features.py:31-35: dict keys,
 * a repeated domain is one feature; [EXT] CRFsuite drops the names it does not know).  0, or -1 with an exception set. */
static int push_gene_features(PyObject *dseq, PyObject *attr_index, buf_t *attr, int64_t *nnz)
{
    const Py_ssize_t nd = PySequence_Fast_GET_SIZE(dseq);
    PyObject *seen_small[16];
    PyObject **seen = nd <= 16 ? seen_small : (PyObject **)malloc(sizeof(PyObject *) * (size_t)nd);
    Py_ssize_t n_seen = 0;
    int failed = seen == NULL;
    if (failed) PyErr_NoMemory();
    for (Py_ssize_t di = 0; di < nd && !failed; ++di) {
        PyObject *name = attr_of(PySequence_Fast_GET_ITEM(dseq, di), s_name);
        if (!name) {
            failed = 1;
            break;
        }
        int dup = 0;
        for (Py_ssize_t k = 0; k < n_seen && !dup; ++k) {
            if (seen[k] == name) {
                dup = 1;
            } else {
                const int eq = PyObject_RichCompareBool(seen[k], name, Py_EQ);
                if (eq < 0) failed = 1;
                dup = eq > 0;
            }
        }
        if (dup || failed) {
            Py_DECREF(name);
            continue;
        }
        seen[n_seen++] = name; /* keeps the reference until the gene is done */
        PyObject *idx = PyDict_GetItemWithError(attr_index, name); /* borrowed */
        if (idx) {
            const long v = PyLong_AsLong(idx);
            if (v == -1 && PyErr_Occurred()) {
                failed = 1;
            } else {
                const int32_t v32 = (int32_t)v;
                if (buf_push(attr, &v32, 4)) failed = 1;
                ++*nnz;
            }
        } else if (PyErr_Occurred()) {
            failed = 1;
        }
    }
    for (Py_ssize_t k = 0; k < n_seen; ++k) Py_DECREF(seen[k]);
    if (seen && seen != seen_small) free(seen);
    return failed ? -1 : 0;
}

static PyObject *pack_protein(PyObject *self, PyObject *args)
{
    PyObject *contigs, *attr_index;
    (void)self;
    if (!PyArg_ParseTuple(args, "OO!", &contigs, &PyDict_Type, &attr_index)) return NULL;
    PyObject *cseq = PySequence_Fast(contigs, "contigs must be a sequence");
    if (!cseq) return NULL;
    buf_t item = {0}, aptr = {0}, attr = {0};
    int64_t n_items = 0, nnz = 0;
    PyObject *result = NULL;
    if (buf_push(&item, &n_items, 8) || buf_push(&aptr, &nnz, 8)) goto done;
    for (Py_ssize_t ci = 0; ci < PySequence_Fast_GET_SIZE(cseq); ++ci) {
        PyObject *gseq = PySequence_Fast(PySequence_Fast_GET_ITEM(cseq, ci), "a contig must be a sequence of genes");
        if (!gseq) goto done;
        for (Py_ssize_t gi = 0; gi < PySequence_Fast_GET_SIZE(gseq); ++gi) {
            PyObject *gene = PySequence_Fast_GET_ITEM(gseq, gi);
            PyObject *prot = attr_of(gene, s_protein);
            if (!prot) {
                Py_DECREF(gseq);
                goto done;
            }
            PyObject *doms = attr_of(prot, s_domains);
            Py_DECREF(prot);
            if (!doms) {
                Py_DECREF(gseq);
                goto done;
            }
            PyObject *dseq = PySequence_Fast(doms, "protein.domains must be a sequence");
            Py_DECREF(doms);
            if (!dseq) {
                Py_DECREF(gseq);
                goto done;
            }
            const int failed = push_gene_features(dseq, attr_index, &attr, &nnz) < 0;
            Py_DECREF(dseq);
            if (failed || buf_push(&aptr, &nnz, 8)) {
                if (!PyErr_Occurred()) PyErr_NoMemory();
                Py_DECREF(gseq);
                goto done;
            }
            ++n_items;
        }
        Py_DECREF(gseq);
        if (buf_push(&item, &n_items, 8)) goto done;
    }
    result = Py_BuildValue("(y#y#y#)", item.p, (Py_ssize_t)item.len, aptr.p, (Py_ssize_t)aptr.len, attr.p ? attr.p : "",
                           (Py_ssize_t)attr.len);
done:
    free(item.p);
    free(aptr.p);
    free(attr.p);
    Py_DECREF(cseq);
    return result;
}

/* a fresh instance of a plain Python class whose __dict__ is `d` (steals d) */
static PyObject *instance_with_dict(PyTypeObject *cls, PyObject *d)
{
    PyObject *o = cls->tp_alloc(cls, 0);
    if (!o) {
        Py_DECREF(d);
        return NULL;
    }
    PyObject **dp = _PyObject_GetDictPtr(o);
    if (dp && !*dp) {
        *dp = d; /* (a fresh instance has no dictionary yet: it takes ours) */
        return o;
    }
    if (PyObject_GenericSetDict(o, d, NULL) < 0) {
        Py_DECREF(d);
        Py_DECREF(o);
        return NULL;
    }
    Py_DECREF(d);
    return o;
}

/* qualifiers.copy() */
static PyObject *copy_mapping(PyObject *q)
{
    if (PyDict_CheckExact(q)) return PyDict_Copy(q);
    return PyObject_CallMethodNoArgs(q, s_copy);
}

static PyObject *annotate_all(PyObject *self, PyObject *args)
{
    PyObject *genes, *probs, *w1, *gene_cls, *prot_cls, *dom_cls;
    (void)self;
    if (!PyArg_ParseTuple(args, "O!OO!OOO", &PyList_Type, &genes, &probs, &PyDict_Type, &w1, &gene_cls, &prot_cls, &dom_cls))
        return NULL;
    const Py_ssize_t n = PyList_GET_SIZE(genes);
    /* probabilities: a list of floats, or a C-contiguous buffer of doubles (a numpy array: the floats are made here) */
    Py_buffer pview;
    const double *pbuf = NULL;
    memset(&pview, 0, sizeof pview);
    if (PyList_Check(probs)) {
        if (PyList_GET_SIZE(probs) != n) {
            PyErr_SetString(PyExc_ValueError, "one probability per gene");
            return NULL;
        }
    } else {
        if (PyObject_GetBuffer(probs, &pview, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) < 0) return NULL;
        if (pview.itemsize != 8 || !pview.format || strcmp(pview.format, "d") != 0 || pview.len != n * 8) {
            PyBuffer_Release(&pview);
            PyErr_SetString(PyExc_ValueError, "one float64 probability per gene");
            return NULL;
        }
        pbuf = (const double *)pview.buf;
    }
    if (!PyType_Check(gene_cls) || !PyType_Check(prot_cls) || (dom_cls != Py_None && !PyType_Check(dom_cls))) {
        if (pbuf) PyBuffer_Release(&pview);
        PyErr_SetString(PyExc_TypeError, "classes expected");
        return NULL;
    }
    PyObject *out = PyList_New(n);
    if (!out) {
        if (pbuf) PyBuffer_Release(&pview);
        return NULL;
    }
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject *gene = PyList_GET_ITEM(genes, i), *p = NULL, *p_owned = NULL;
        if (pbuf) {
            p = p_owned = PyFloat_FromDouble(pbuf[i]);
            if (!p) {
                PyBuffer_Release(&pview);
                Py_DECREF(out);
                return NULL;
            }
        } else {
            p = PyList_GET_ITEM(probs, i);
        }
        PyObject *gd = NULL, *pd = NULL, *new_doms = NULL, *npd = NULL, *np_ = NULL, *ngd = NULL, *q = NULL, *ng = NULL;
        int other = 0; /* an object of another class: give up, the caller takes the generic route */
        if ((PyObject *)Py_TYPE(gene) != gene_cls) {
            other = 1;
            goto next;
        }
        gd = PyObject_GenericGetDict(gene, NULL);
        if (!gd) goto fail;
        PyObject *prot = PyDict_GetItemWithError(gd, s_protein); /* borrowed */
        if (!prot || (PyObject *)Py_TYPE(prot) != prot_cls) {
            if (PyErr_Occurred()) goto fail;
            other = 1;
            goto next;
        }
        pd = PyObject_GenericGetDict(prot, NULL);
        if (!pd) goto fail;
        PyObject *doms = PyDict_GetItemWithError(pd, s_domains); /* borrowed */
        if (!doms || !PyList_CheckExact(doms)) {
            if (PyErr_Occurred()) goto fail;
            other = 1;
            goto next;
        }
        const Py_ssize_t nd = PyList_GET_SIZE(doms);
        new_doms = PyList_New(nd);
        if (!new_doms) goto fail;
        for (Py_ssize_t j = 0; j < nd; ++j) {
            PyObject *d = PyList_GET_ITEM(doms, j);
            if ((PyObject *)Py_TYPE(d) != dom_cls) {
                other = 1;
                goto next;
            }
            PyObject *dd = PyObject_GenericGetDict(d, NULL);
            if (!dd) goto fail;
            PyObject *ndd = PyDict_Copy(dd);
            Py_DECREF(dd);
            if (!ndd) goto fail;
            PyObject *name = PyDict_GetItemWithError(ndd, s_name);     /* borrowed */
            PyObject *dq = PyDict_GetItemWithError(ndd, s_qualifiers); /* borrowed */
            if (!name || !dq) {
                Py_DECREF(ndd);
                if (PyErr_Occurred()) goto fail;
                other = 1;
                goto next;
            }
            PyObject *w = PyDict_GetItemWithError(w1, name); /* borrowed; absent: None (crf/__init__.py:264) */
            if (!w && PyErr_Occurred()) {
                Py_DECREF(ndd);
                goto fail;
            }
            PyObject *nq = copy_mapping(dq);
            if (!nq || PyDict_SetItem(ndd, s_probability, p) < 0 || PyDict_SetItem(ndd, s_cluster_weight, w ? w : Py_None) < 0 ||
                PyDict_SetItem(ndd, s_qualifiers, nq) < 0) {
                Py_XDECREF(nq);
                Py_DECREF(ndd);
                goto fail;
            }
            Py_DECREF(nq);
            PyObject *ndo = instance_with_dict((PyTypeObject *)dom_cls, ndd);
            if (!ndo) goto fail;
            PyList_SET_ITEM(new_doms, j, ndo);
        }
        npd = PyDict_Copy(pd);
        if (!npd || PyDict_SetItem(npd, s_domains, new_doms) < 0) goto fail;
        np_ = instance_with_dict((PyTypeObject *)prot_cls, npd);
        npd = NULL;
        if (!np_) goto fail;
        ngd = PyDict_Copy(gd);
        if (!ngd) goto fail;
        PyObject *gq = PyDict_GetItemWithError(ngd, s_qualifiers); /* borrowed */
        if (!gq) {
            if (PyErr_Occurred()) goto fail;
            other = 1;
            goto next;
        }
        q = copy_mapping(gq);
        if (!q || PyDict_SetItem(ngd, s_protein, np_) < 0 || PyDict_SetItem(ngd, s_qualifiers, q) < 0 ||
            PyDict_SetItem(ngd, s__probability, p) < 0)
            goto fail;
        ng = instance_with_dict((PyTypeObject *)gene_cls, ngd);
        ngd = NULL;
        if (!ng) goto fail;
        PyList_SET_ITEM(out, i, ng);
        ng = NULL;
    next:
        Py_XDECREF(p_owned);
        Py_XDECREF(gd);
        Py_XDECREF(pd);
        Py_XDECREF(new_doms);
        Py_XDECREF(npd);
        Py_XDECREF(np_);
        Py_XDECREF(ngd);
        Py_XDECREF(q);
        if (other) {
            if (pbuf) PyBuffer_Release(&pview);
            Py_DECREF(out);
            Py_RETURN_NONE;
        }
        continue;
    fail:
        Py_XDECREF(p_owned);
        if (pbuf) PyBuffer_Release(&pview);
        Py_XDECREF(gd);
        Py_XDECREF(pd);
        Py_XDECREF(new_doms);
        Py_XDECREF(npd);
        Py_XDECREF(np_);
        Py_XDECREF(ngd);
        Py_XDECREF(q);
        Py_XDECREF(ng);
        Py_DECREF(out);
        return NULL;
    }
    if (pbuf) PyBuffer_Release(&pview);
    return out;
}

/* sort_group(genes, start_key[, attr_index]) -> (genes as a list, contigs as a list of lists[, item_ptr, attr_ptr, attr_id]) or None.
 * gecco/crf/__init__.py:199-206: genes sorted by (source.id, start), every gene's domain list sorted by start IN PLACE, genes
 * grouped by source.id.  Annotation pipelines emit genes in that order already, and sorted() is stable: when the input is
 * non-decreasing in (source.id, start) the sorted list IS the input, and only the (rare) unsorted domain lists are sorted
 * (list.sort(key=start_key), the reference's own call).  Anything else -- an unsorted input, keys that do not compare --
 * returns None and the caller runs the Python statements.  With `attr_index` the same pass also packs the protein features
 * (pack_protein's arrays): every object is then visited once while it is in cache -- on a metagenome the loops over the
 * objects are memory-latency bound, and three passes cost three times the misses. */
static PyObject *s_source, *s_id, *s_start, *s_sort, *s_key;
static PyObject *sort_group(PyObject *self, PyObject *args)
{
    PyObject *genes, *start_key, *attr_index = NULL;
    (void)self;
    if (!PyArg_ParseTuple(args, "OO|O!", &genes, &start_key, &PyDict_Type, &attr_index)) return NULL;
    PyObject *seq = PySequence_Fast(genes, "genes must be iterable");
    if (!seq) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    PyObject *out_genes = PyList_New(n), *contigs = PyList_New(0), *cur = NULL, *prev_id = NULL, *prev_start = NULL, *result = NULL;
    PyObject *prev_src = NULL; /* owned: an object kept alive cannot lend its address to the next gene's source */
    buf_t item = {0}, aptr = {0}, attr = {0};
    int64_t n_items = 0, nnz = 0;
    int sorted_input = 1;
    if (!out_genes || !contigs) goto done;
    if (attr_index && (buf_push(&item, &n_items, 8) || buf_push(&aptr, &nnz, 8))) goto done;
    for (Py_ssize_t i = 0; i < n && sorted_input; ++i) {
        PyObject *g = PySequence_Fast_GET_ITEM(seq, i);
        Py_INCREF(g);
        PyList_SET_ITEM(out_genes, i, g);
        PyObject *src = attr_of(g, s_source);
        if (!src) goto done;
        int same = 0;
        PyObject *id = NULL;
        if (src == prev_src && prev_id) {  /* the same source object as the gene before: the same id, no lookup */
            same = 1;
            id = prev_id;
            Py_INCREF(id);
        } else {
            id = attr_of(src, s_id);
        }
        PyObject *start = id ? attr_of(g, s_start) : NULL;
        if (!start) {
            Py_DECREF(src);
            Py_XDECREF(id);
            goto done;
        }
        Py_XDECREF(prev_src);
        prev_src = src; /* (takes the reference) */
        if (prev_id && !same) {
            const int lt = PyObject_RichCompareBool(prev_id, id, Py_LT);
            same = lt == 0 ? PyObject_RichCompareBool(prev_id, id, Py_EQ) : 0;
            if (lt < 0 || same < 0) {
                Py_DECREF(id);
                Py_DECREF(start);
                goto done;
            }
            if (!lt && !same) sorted_input = 0;
        }
        if (same) {
            const int le = key_cmp(prev_start, start, Py_LE);
            if (le < 0) {
                Py_DECREF(id);
                Py_DECREF(start);
                goto done;
            }
            if (!le) sorted_input = 0;
        }
        if (!same) {  /* a new contig */
            if (attr_index && cur && buf_push(&item, &n_items, 8)) {
                Py_DECREF(id);
                Py_DECREF(start);
                goto done;
            }
            cur = PyList_New(0);
            if (!cur || PyList_Append(contigs, cur) < 0) {
                Py_XDECREF(cur);
                cur = NULL;
                Py_DECREF(id);
                Py_DECREF(start);
                goto done;
            }
            Py_DECREF(cur);  /* (the list of contigs holds it) */
        }
        if (PyList_Append(cur, g) < 0) {
            Py_DECREF(id);
            Py_DECREF(start);
            goto done;
        }
        Py_XDECREF(prev_id);
        Py_XDECREF(prev_start);
        prev_id = id;
        prev_start = start;
        /* the gene's domains by start, in place */
        PyObject *prot = attr_of(g, s_protein);
        PyObject *doms = prot ? attr_of(prot, s_domains) : NULL;
        Py_XDECREF(prot);
        if (!doms) goto done;
        if (!PyList_CheckExact(doms)) {  /* (the reference calls .sort on whatever it is: leave that to Python) */
            Py_DECREF(doms);
            sorted_input = 0;
            break;
        }
        int dom_sorted = 1;
        PyObject *ps = NULL;
        for (Py_ssize_t j = 0; j < PyList_GET_SIZE(doms) && dom_sorted > 0; ++j) {
            PyObject *ds = attr_of(PyList_GET_ITEM(doms, j), s_start);
            if (!ds) {
                dom_sorted = -1;
                break;
            }
            if (ps) dom_sorted = key_cmp(ps, ds, Py_LE);
            Py_XDECREF(ps);
            ps = ds;
        }
        Py_XDECREF(ps);
        if (dom_sorted < 0) {
            Py_DECREF(doms);
            goto done;
        }
        if (!dom_sorted) {
            PyObject *meth = PyObject_GetAttr(doms, s_sort), *kw = PyDict_New(), *empty = PyTuple_New(0), *r = NULL;
            if (meth && kw && empty && PyDict_SetItem(kw, s_key, start_key) == 0) r = PyObject_Call(meth, empty, kw);
            Py_XDECREF(meth);
            Py_XDECREF(kw);
            Py_XDECREF(empty);
            if (!r) {
                Py_DECREF(doms);
                goto done;
            }
            Py_DECREF(r);
        }
        if (attr_index) {
            if (push_gene_features(doms, attr_index, &attr, &nnz) < 0 || buf_push(&aptr, &nnz, 8)) {
                Py_DECREF(doms);
                goto done;
            }
            ++n_items;
        }
        Py_DECREF(doms);
    }
    if (!sorted_input) {
        result = Py_None;
        Py_INCREF(result);
    } else if (attr_index) {
        if (cur && buf_push(&item, &n_items, 8)) goto done;
        result = Py_BuildValue("(OOy#y#y#)", out_genes, contigs, item.p, (Py_ssize_t)item.len, aptr.p, (Py_ssize_t)aptr.len,
                               attr.p ? attr.p : "", (Py_ssize_t)attr.len);
    } else {
        result = PyTuple_Pack(2, out_genes, contigs);
    }
done:
    free(item.p);
    free(aptr.p);
    free(attr.p);
    Py_XDECREF(prev_src);
    Py_XDECREF(prev_id);
    Py_XDECREF(prev_start);
    Py_XDECREF(out_genes);
    Py_XDECREF(contigs);
    Py_DECREF(seq);
    return result;
}

static PyMethodDef methods[] = {
    {"sort_group", sort_group, METH_VARARGS, "sort_group(genes, start_key[, attr_index]) -> (genes, contigs[, item_ptr, attr_ptr, attr_id]) when the input is in (source.id, start) order, else None"},
    {"pack_protein", pack_protein, METH_VARARGS, "pack_protein(contigs, attr_index) -> (item_ptr, attr_ptr, attr_id) as bytes"},
    {"annotate_all", annotate_all, METH_VARARGS, "annotate_all(genes, probs, w1, gene_cls, prot_cls, dom_cls) -> list or None"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_objpath", "object-model loops of gecco_amd.crf.ClusterCRF", -1, methods,
                                       NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__objpath(void)
{
    s_protein = PyUnicode_InternFromString("protein");
    s_domains = PyUnicode_InternFromString("domains");
    s_name = PyUnicode_InternFromString("name");
    s_probability = PyUnicode_InternFromString("probability");
    s_cluster_weight = PyUnicode_InternFromString("cluster_weight");
    s_qualifiers = PyUnicode_InternFromString("qualifiers");
    s__probability = PyUnicode_InternFromString("_probability");
    s_copy = PyUnicode_InternFromString("copy");
    s_source = PyUnicode_InternFromString("source");
    s_id = PyUnicode_InternFromString("id");
    s_start = PyUnicode_InternFromString("start");
    s_sort = PyUnicode_InternFromString("sort");
    s_key = PyUnicode_InternFromString("key");
    if (!s_source || !s_id || !s_start || !s_sort || !s_key) return NULL;
    if (!s_protein || !s_domains || !s_name || !s_probability || !s_cluster_weight || !s_qualifiers || !s__probability || !s_copy)
        return NULL;
    return PyModule_Create(&moduledef);
}
