/* Host-side helper for the replay packer (CPython C API, built with gcc; optional).
 *
 * `pack_replay` has to hand the C packer one data pointer per (state, field): 16 384 states x 9 arrays per PPO
 * iteration.  Extracting them in a Python loop costs as much as the packing itself (~0.08 s per iteration); this
 * walks the list through the buffer protocol in C instead.  It is glue, not part of the hot path: packer.py falls
 * back to the Python loop when the module is missing, and the C ABI of libupamd.so does not depend on it.
 *
 *   addr_table(states, ptrs: uint64[9, T], pad_n: int32[T], pad_e: int32[T], node_dim, numerical_dim) -> int
 *
 * fills the tables for every state whose 9 fields are C-contiguous buffers of the wire dtypes
 * (urban_planning/envs/observation_extractor.py:207-228: f32, f32, i64, f32, bool x4, f32) and returns -1; at the
 * first state that is anything else it returns that state's index (the caller then takes the slow path).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

static const char KIND[9] = {'f', 'f', 'q', 'f', '?', '?', '?', '?', 'f'};
static const int ITEM[9] = {4, 4, 8, 4, 1, 1, 1, 1, 4};

static int format_ok(const char *fmt, int field) {
    if (!fmt) return 0;
    while (*fmt == '@' || *fmt == '=' || *fmt == '<') ++fmt;
    if (fmt[0] == '\0' || fmt[1] != '\0') return 0;
    if (KIND[field] == 'q') return fmt[0] == 'q' || fmt[0] == 'l';
    return fmt[0] == KIND[field];
}

static PyObject *addr_table(PyObject *self, PyObject *args) {
    PyObject *states, *ptrs_o, *padn_o, *pade_o;
    int node_dim, numerical_dim;
    (void)self;
    if (!PyArg_ParseTuple(args, "OOOOii", &states, &ptrs_o, &padn_o, &pade_o, &node_dim, &numerical_dim)) return NULL;
    PyObject *seq = PySequence_Fast(states, "states must be a sequence");
    if (!seq) return NULL;
    const Py_ssize_t T = PySequence_Fast_GET_SIZE(seq);
    Py_buffer bp, bn, be;
    if (PyObject_GetBuffer(ptrs_o, &bp, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) < 0) { Py_DECREF(seq); return NULL; }
    if (PyObject_GetBuffer(padn_o, &bn, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) < 0) { PyBuffer_Release(&bp); Py_DECREF(seq); return NULL; }
    if (PyObject_GetBuffer(pade_o, &be, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) < 0) {
        PyBuffer_Release(&bp); PyBuffer_Release(&bn); Py_DECREF(seq); return NULL;
    }
    long bad = -1;
    if (bp.len < (Py_ssize_t)(9 * T * 8) || bn.len < (Py_ssize_t)(T * 4) || be.len < (Py_ssize_t)(T * 4)) {
        PyErr_SetString(PyExc_ValueError, "addr_table: output tables too small");
        bad = -2;
    }
    uint64_t *ptrs = (uint64_t *)bp.buf;
    int32_t *pad_n = (int32_t *)bn.buf, *pad_e = (int32_t *)be.buf;
    for (Py_ssize_t t = 0; t < T && bad == -1; ++t) {
        PyObject *st = PySequence_Fast(PySequence_Fast_GET_ITEM(seq, t), "state must be a sequence");
        if (!st) { PyErr_Clear(); bad = (long)t; break; }
        if (PySequence_Fast_GET_SIZE(st) != 9) { Py_DECREF(st); bad = (long)t; break; }
        Py_ssize_t n_rows = -1, e_rows = -1;
        for (int f = 0; f < 9; ++f) {
            Py_buffer v;
            if (PyObject_GetBuffer(PySequence_Fast_GET_ITEM(st, f), &v, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) < 0) {
                PyErr_Clear();
                bad = (long)t;
                break;
            }
            int ok = v.itemsize == ITEM[f] && format_ok(v.format, f) && v.ndim >= 1;
            if (ok) {
                const Py_ssize_t rows = v.shape[0];
                if (f == 1) { n_rows = rows; ok = v.ndim == 2 && v.shape[1] == node_dim; }
                else if (f == 2) { e_rows = rows; ok = v.ndim == 2 && v.shape[1] == 2; }
                /* the C packer copies numerical_dim / node_dim / 3 floats out of these three: a short array would be
                 * an out-of-bounds host read, so anything else goes to the validating slow path (which raises) */
                else if (f == 0) ok = v.len / v.itemsize == numerical_dim;
                else if (f == 3) ok = v.len / v.itemsize == node_dim;
                else if (f == 8) ok = v.len / v.itemsize == 3;
                else if (f == 4 || f == 7) ok = rows == n_rows;
                else if (f == 5 || f == 6) ok = rows == e_rows;
            }
            if (ok) ptrs[(Py_ssize_t)f * T + t] = (uint64_t)(uintptr_t)v.buf;
            PyBuffer_Release(&v);          /* the array outlives the call: the caller keeps `states` alive */
            if (!ok) { bad = (long)t; break; }
        }
        Py_DECREF(st);
        if (bad == -1) {
            if (n_rows > INT32_MAX || e_rows > INT32_MAX) { bad = (long)t; break; }
            pad_n[t] = (int32_t)n_rows;
            pad_e[t] = (int32_t)e_rows;
        }
    }
    PyBuffer_Release(&bp);
    PyBuffer_Release(&bn);
    PyBuffer_Release(&be);
    Py_DECREF(seq);
    if (bad == -2) return NULL;
    return PyLong_FromLong(bad);
}

static PyMethodDef METHODS[] = {{"addr_table", addr_table, METH_VARARGS, "fill the packer's pointer / pad tables"},
                                {NULL, NULL, 0, NULL}};
static struct PyModuleDef MODULE = {PyModuleDef_HEAD_INIT, "_upamd_host", "host-side helpers of the replay packer", -1, METHODS,
                                    NULL, NULL, NULL, NULL};
PyMODINIT_FUNC PyInit__upamd_host(void) { return PyModule_Create(&MODULE); }
