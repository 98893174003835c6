/* q1rows.c - host-side helper of the drop-in Python surface (q1physrl_amd/env.py): RLlib hands VectorPhysEnv.vector_step a list of N
 * tuples whose members are Python ints, floats, NumPy scalars or arrays of length >= 1, and the reference's ActionDecoder._fix_actions
 * (env.py:221-223) turns them into an (N, A) float64 array with a per-element Python loop - 86 % of its wall time at N = 64 k
 * (SURVEY.md 8a row a2).  This is that loop in C against the CPython API: one pass, no temporaries, the first element of a buffer-
 * protocol object read in place.  Anything it does not recognise makes it return None and env.py falls back to its NumPy
 * formulations (which reproduce the reference's errors).  A CPython extension module: built by q1physrl_amd/build.py with gcc;
 * optional - without it the NumPy path runs.
 *
 *   rows(actions, width, out) -> True | None      out: writable C-contiguous float64 buffer of len(actions) * width elements
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>
#include <stdint.h>
#include <string.h>

/* first element of a NumPy array (the common case: RLlib's Box component arrives as a float32 array of shape (1,)) read through
 * the array struct - PyObject_GetBuffer on an ndarray costs ~200 ns, this a few */
static int first_of_ndarray(PyObject* x, double* out) {
    PyArrayObject* a = (PyArrayObject*)x;
    if (PyArray_SIZE(a) < 1 || !PyArray_ISALIGNED(a) || !PyArray_ISNOTSWAPPED(a)) return -1;
    const void* p = PyArray_DATA(a);
    switch (PyArray_TYPE(a)) {
        case NPY_FLOAT32: *out = (double)*(const float*)p; return 0;
        case NPY_FLOAT64: *out = *(const double*)p; return 0;
        case NPY_INT64: *out = (double)*(const int64_t*)p; return 0;
        case NPY_INT32: *out = (double)*(const int32_t*)p; return 0;
        case NPY_BOOL: case NPY_UINT8: *out = (double)*(const uint8_t*)p; return 0;
        case NPY_INT8: *out = (double)*(const int8_t*)p; return 0;
        default: return -2;                                              /* other dtypes: the buffer route below */
    }
}

/* first element of a buffer-protocol object as double; 0 on success */
static int first_of_buffer(PyObject* x, double* out) {
    Py_buffer v;
    if (PyObject_GetBuffer(x, &v, PyBUF_FORMAT | PyBUF_STRIDES) != 0) { PyErr_Clear(); return -1; }
    int ok = -1;
    if (v.len >= v.itemsize && v.itemsize > 0 && v.buf) {
        const char* f = v.format ? v.format : "B";
        while (*f == '@' || *f == '=' || *f == '<') ++f;                 /* native / little-endian prefixes */
        const void* p = v.buf;
        ok = 0;
        switch (*f) {
            case 'd': { double t; memcpy(&t, p, 8); *out = t; break; }
            case 'f': { float t; memcpy(&t, p, 4); *out = (double)t; break; }
            case 'b': *out = (double)*(const int8_t*)p; break;
            case 'B': case '?': *out = (double)*(const uint8_t*)p; break;
            case 'h': { int16_t t; memcpy(&t, p, 2); *out = (double)t; break; }
            case 'H': { uint16_t t; memcpy(&t, p, 2); *out = (double)t; break; }
            case 'i': { int32_t t; memcpy(&t, p, 4); *out = (double)t; break; }
            case 'I': { uint32_t t; memcpy(&t, p, 4); *out = (double)t; break; }
            case 'l': case 'q': if (v.itemsize == 8) { int64_t t; memcpy(&t, p, 8); *out = (double)t; } else { int32_t t; memcpy(&t, p, 4); *out = (double)t; } break;
            case 'L': case 'Q': if (v.itemsize == 8) { uint64_t t; memcpy(&t, p, 8); *out = (double)t; } else { uint32_t t; memcpy(&t, p, 4); *out = (double)t; } break;
            default: ok = -1;
        }
    }
    PyBuffer_Release(&v);
    return ok;
}

static int component(PyObject* x, double* out) {
    if (PyFloat_CheckExact(x)) { *out = PyFloat_AS_DOUBLE(x); return 0; }
    if (PyLong_CheckExact(x) || PyBool_Check(x)) {
        const double d = PyLong_AsDouble(x);
        if (d == -1.0 && PyErr_Occurred()) { PyErr_Clear(); return -1; }
        *out = d;
        return 0;
    }
    if (PyArray_Check(x)) {
        const int r = first_of_ndarray(x, out);
        if (r != -2) return r;
    }
    /* other ndarrays of length >= 1 and NumPy scalars ONLY: bytes / bytearray / memoryview also speak the buffer protocol, but the
     * reference's np.ravel(x)[0] (env.py:223) yields a non-numeric element for them and fails later - they go back to the NumPy path */
    if ((PyArray_Check(x) || PyArray_IsScalar(x, Generic)) && PyObject_CheckBuffer(x)) return first_of_buffer(x, out);
    if (PyFloat_Check(x) || PyLong_Check(x)) {                           /* subclasses */
        const double d = PyFloat_AsDouble(x);
        if (d == -1.0 && PyErr_Occurred()) { PyErr_Clear(); return -1; }
        *out = d;
        return 0;
    }
    return -1;                                                           /* nested lists etc.: the caller's NumPy path decides */
}

static PyObject* rows(PyObject* self, PyObject* args) {
    PyObject *actions, *outobj;
    Py_ssize_t width;
    (void)self;
    if (!PyArg_ParseTuple(args, "OnO", &actions, &width, &outobj)) return NULL;
    if (!(PyList_CheckExact(actions) || PyTuple_CheckExact(actions)) || width <= 0) Py_RETURN_NONE;
    Py_buffer ob;
    if (PyObject_GetBuffer(outobj, &ob, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(actions);
    int ok = ob.itemsize == 8 && ob.format && ob.format[strlen(ob.format) - 1] == 'd' && ob.len == n * width * 8;
    double* dst = (double*)ob.buf;
    PyObject** items = PySequence_Fast_ITEMS(actions);
    for (Py_ssize_t i = 0; ok && i < n; ++i) {
        PyObject* row = items[i];
        if (!(PyTuple_CheckExact(row) || PyList_CheckExact(row)) || PySequence_Fast_GET_SIZE(row) != width) { ok = 0; break; }
        PyObject** comp = PySequence_Fast_ITEMS(row);
        for (Py_ssize_t j = 0; j < width; ++j)
            if (component(comp[j], dst + i * width + j) != 0) { ok = 0; break; }
    }
    PyBuffer_Release(&ob);
    if (ok) Py_RETURN_TRUE;
    Py_RETURN_NONE;
}

static PyMethodDef methods[] = {
    {"rows", rows, METH_VARARGS, "rows(actions, width, out_float64) -> True, or None when the input is not a list of width-tuples of scalars / buffers"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_q1rows", "fast list-of-tuples -> float64 rows", -1, methods, NULL, NULL, NULL, NULL};
PyMODINIT_FUNC PyInit__q1rows(void) {
    import_array();
    return PyModule_Create(&moduledef);
}
