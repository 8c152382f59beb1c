/* CPython glue of the host layer: walks Python-side molecule descriptions and fills the C-ABI descriptor arrays of
 * include/nvmolkit_amd.h (nvmk_flat_molecule, nvmk_host_terms) with pointers INTO the caller's numpy arrays — no copies, no
 * per-array Python calls.  It is to the Python modules of nvmolkit_amd what the Boost.Python list walkers are to the reference's bindings
 * (nvmolkit/embedMolecules.cpp, mmffOptimization.cpp: boost::python::extract over the molecule list); the product library
 * itself stays free of Python.  Built by nvmolkit_amd/_build.py into lib/_nvmk_pyglue.so, loaded with ctypes.PyDLL (GIL held).
 *
 * Fast path: an array that is C-contiguous with dtype int32 / int64 (indices) or float64 (parameters) is referenced where it
 * lies.  Anything else (lists, other dtypes, strided views) goes through the `convert(obj, is_par)` callable the caller
 * passes — numpy.ascontiguousarray — and the converted array is appended to `keep`, which the caller holds for as long as
 * the descriptors are in use.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

#include "nvmolkit_amd.h"

/* Where numpy's headers are at hand (NVMK_GLUE_NUMPY, set by _build.py) an ndarray is read through numpy's C API — a handful
 * of struct fields — instead of the buffer protocol, whose PyBUF_FORMAT request makes numpy build a format string per call:
 * 18 such calls per molecule were 5 of the 6.4 us the walk took.  The API table is fetched on first use (the glue is loaded
 * with ctypes.PyDLL, not imported); if that fails the buffer protocol below serves everything, as it does without the headers. */
#ifdef NVMK_GLUE_NUMPY
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>
static int g_numpy = 0; /* 0: not tried, 1: usable, -1: not available */
static int numpy_ready(void) {
  if (g_numpy == 0) {
    g_numpy = _import_array() == 0 ? 1 : -1;
    if (g_numpy < 0) PyErr_Clear();
  }
  return g_numpy > 0;
}
#endif

static int buffer_of(PyObject* obj, int is_par, PyObject* keep, PyObject* convert, const void** ptr, Py_ssize_t* len, int* itemsize) {
#ifdef NVMK_GLUE_NUMPY
  if (numpy_ready() && PyArray_Check(obj)) {
    PyArrayObject* arr = (PyArrayObject*)obj;
    const int      t   = PyArray_TYPE(arr);
    const int      sz  = (int)PyArray_ITEMSIZE(arr);
    const int      ok  = is_par ? (t == NPY_DOUBLE) : ((t == NPY_INT || t == NPY_LONG || t == NPY_LONGLONG) && (sz == 4 || sz == 8));
    if (ok && PyArray_IS_C_CONTIGUOUS(arr) && PyArray_ISNOTSWAPPED(arr) && PyArray_ISALIGNED(arr)) {
      *ptr      = PyArray_DATA(arr);
      *len      = (Py_ssize_t)PyArray_NBYTES(arr);
      *itemsize = sz;
      return 0;
    }
  }
#endif
  Py_buffer view;
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (PyObject_GetBuffer(obj, &view, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) == 0) {
      const char* f = view.format ? view.format : "B";
      while (*f == '@' || *f == '=' || *f == '<') ++f;
      int ok;
      if (is_par) {
        ok = (f[0] == 'd' && f[1] == '\0' && view.itemsize == 8);
      } else {
        ok = ((f[0] == 'i' || f[0] == 'l' || f[0] == 'q') && f[1] == '\0' && (view.itemsize == 4 || view.itemsize == 8));
      }
      if (ok || view.len == 0) {
        *ptr      = view.buf;
        *len      = view.len;
        *itemsize = ok ? (int)view.itemsize : (is_par ? 8 : 4);
        PyBuffer_Release(&view); /* the object outlives the descriptors (caller's list / keep) */
        return 0;
      }
      PyBuffer_Release(&view);
    } else {
      PyErr_Clear();
    }
    if (attempt == 1) break;
    PyObject* conv = PyObject_CallFunction(convert, "Oi", obj, is_par);
    if (conv == NULL) return -1;
    if (PyList_Append(keep, conv) != 0) {
      Py_DECREF(conv);
      return -1;
    }
    Py_DECREF(conv); /* keep holds it */
    obj = conv;
  }
  PyErr_SetString(PyExc_TypeError, "term arrays must convert to C-contiguous int32 / int64 (indices) or float64 (parameters)");
  return -1;
}

/* pair = (idx, par) -> *out */
static int terms_of(PyObject* pair, int n_idx, int n_par, PyObject* keep, PyObject* convert, nvmk_host_terms* out, const char* what,
                    Py_ssize_t m, int g) {
  PyObject* fast = PySequence_Fast(pair, "a term group must be an (idx, par) pair");
  if (fast == NULL) return -1;
  if (PySequence_Fast_GET_SIZE(fast) != 2) {
    Py_DECREF(fast);
    PyErr_Format(PyExc_ValueError, "%s: molecule %zd, group %d: expected an (idx, par) pair", what, m, g);
    return -1;
  }
  const void* ip  = NULL;
  const void* pp  = NULL;
  Py_ssize_t  il = 0, pl = 0;
  int         isz = 4, psz = 8;
  int         rc = buffer_of(PySequence_Fast_GET_ITEM(fast, 0), 0, keep, convert, &ip, &il, &isz);
  if (rc == 0 && n_par > 0) rc = buffer_of(PySequence_Fast_GET_ITEM(fast, 1), 1, keep, convert, &pp, &pl, &psz);
  /* the descriptors point into the arrays this pair owns: the pair stays alive with `keep` (an (idx, par) pair that a
   * property or an iterator made on the fly would otherwise be gone before the build reads its arrays) */
  if (rc == 0 && PyList_Append(keep, fast) != 0) rc = -1;
  Py_DECREF(fast);
  if (rc != 0) return -1;
  const Py_ssize_t row = (Py_ssize_t)isz * n_idx;
  if (il % row != 0) {
    PyErr_Format(PyExc_ValueError, "%s: molecule %zd, group %d: idx does not hold whole rows of %d indices", what, m, g, n_idx);
    return -1;
  }
  const Py_ssize_t n = il / row;
  if (n_par > 0 && pl != n * n_par * 8) {
    PyErr_Format(PyExc_ValueError, "%s: molecule %zd, group %d: %zd index rows but %zd parameter values (%d per row)", what, m, g, n,
                 pl / 8, n_par);
    return -1;
  }
  if (n >= INT32_MAX) {
    PyErr_Format(PyExc_ValueError, "%s: molecule %zd, group %d: too many rows", what, m, g);
    return -1;
  }
  out->n_terms   = (int32_t)n;
  out->idx_bytes = isz;
  out->idx       = ip;
  out->par       = (const double*)pp;
  return 0;
}

static int groups_of(PyObject* seq, const int32_t* n_idx, const int32_t* n_par, int n_groups, PyObject* keep, PyObject* convert,
                     nvmk_host_terms* out, const char* what, Py_ssize_t m) {
  PyObject* fast = PySequence_Fast(seq, "term groups must be a sequence of (idx, par) pairs");
  if (fast == NULL) return -1;
  if (PySequence_Fast_GET_SIZE(fast) != n_groups) {
    PyErr_Format(PyExc_ValueError, "%s: molecule %zd: needs %d term groups, got %zd", what, m, n_groups, PySequence_Fast_GET_SIZE(fast));
    Py_DECREF(fast);
    return -1;
  }
  for (int g = 0; g < n_groups; ++g) {
    if (terms_of(PySequence_Fast_GET_ITEM(fast, g), n_idx[g], n_par[g], keep, convert, &out[g], what, m, g) != 0) {
      Py_DECREF(fast);
      return -1;
    }
  }
  const int kept = PyList_Append(keep, fast);  /* (as above: the sequence of pairs, if it was made for this call) */
  Py_DECREF(fast);
  return kept;
}

static const int32_t DG_IDX[3] = {2, 4, 1}, DG_PAR[3] = {3, 2, 0};
static const int32_t ETK_IDX[6] = {4, 4, 2, 2, 3, 2}, ETK_PAR[6] = {12, 4, 4, 4, 2, 4};

/* mols: sequence of objects with attributes n_atoms, dg, etk (or None), checks, num_impropers (embedMolecules.FlatMolecule).
 * out: n descriptors; check_kind / check_idx / check_par: caller-allocated arrays for `check_capacity` checks in all, filled
 * here and referenced by the descriptors.  Returns the number of checks written, -1 with a Python exception set. */
int64_t nvmk_py_gather_flat_molecules(PyObject* mols, nvmk_flat_molecule* out, int32_t* check_kind, int32_t* check_idx, double* check_par,
                                      int64_t check_capacity, PyObject* keep, PyObject* convert) {
  PyObject* fast = PySequence_Fast(mols, "molecules must be a sequence");
  if (fast == NULL) return -1;
  const Py_ssize_t n    = PySequence_Fast_GET_SIZE(fast);
  int64_t          used = 0;
  for (Py_ssize_t m = 0; m < n; ++m) {
    PyObject*           mol = PySequence_Fast_GET_ITEM(fast, m);
    nvmk_flat_molecule* d   = &out[m];
    memset(d, 0, sizeof(*d));
    PyObject* a = PyObject_GetAttrString(mol, "n_atoms");
    if (a == NULL) goto fail;
    d->n_atoms = (int32_t)PyLong_AsLong(a);
    Py_DECREF(a);
    a = PyObject_GetAttrString(mol, "num_impropers");
    if (a == NULL) goto fail;
    d->num_impropers = (int32_t)PyLong_AsLong(a);
    Py_DECREF(a);
    if (PyErr_Occurred()) goto fail;
    a = PyObject_GetAttrString(mol, "dg");
    if (a == NULL) goto fail;
    int rc = groups_of(a, DG_IDX, DG_PAR, 3, keep, convert, d->dg, "FlatMolecule.dg", m);
    Py_DECREF(a);
    if (rc != 0) goto fail;
    a = PyObject_GetAttrString(mol, "etk");
    if (a == NULL) goto fail;
    if (a != Py_None) {
      d->has_etk = 1;
      rc         = groups_of(a, ETK_IDX, ETK_PAR, 6, keep, convert, d->etk, "FlatMolecule.etk", m);
    }
    Py_DECREF(a);
    if (rc != 0) goto fail;
    a = PyObject_GetAttrString(mol, "checks");
    if (a == NULL) goto fail;
    if (PyObject_HasAttrString(a, "kind") && PyObject_HasAttrString(a, "idx") && PyObject_HasAttrString(a, "par")) {
      /* embedMolecules.StereoChecks: kind (n,) int32, idx (n, 5) int32, par (n, 2) float64 — three copies, no tuple walk */
      PyObject*  parts[3] = {PyObject_GetAttrString(a, "kind"), PyObject_GetAttrString(a, "idx"), PyObject_GetAttrString(a, "par")};
      Py_buffer  view[3];
      int        got = 0, bad = 0;
      for (int k = 0; k < 3 && !bad; ++k) {
        if (parts[k] == NULL || PyObject_GetBuffer(parts[k], &view[k], PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) {
          bad = 1;
        } else {
          ++got;
          const char* f = view[k].format ? view[k].format : "B";
          while (*f == '@' || *f == '=' || *f == '<') ++f;
          if (k < 2 ? !(f[0] == 'i' && f[1] == 0 && view[k].itemsize == 4) : !(f[0] == 'd' && f[1] == 0 && view[k].itemsize == 8)) bad = 1;
        }
      }
      Py_ssize_t nc = bad ? 0 : view[0].len / 4;
      if (!bad && (view[1].len != nc * 20 || view[2].len != nc * 16)) bad = 1;
      if (!bad) { /* the descriptor points INTO the molecule's own arrays (no copy); `keep` holds the object that owns them */
        d->n_checks   = (int32_t)nc;
        d->check_kind = (const int32_t*)view[0].buf;
        d->check_idx  = (const int32_t*)view[1].buf;
        d->check_par  = (const double*)view[2].buf;
        if (PyList_Append(keep, a) != 0) bad = 1;
      }
      for (int k = 0; k < got; ++k) PyBuffer_Release(&view[k]);
      for (int k = 0; k < 3; ++k) Py_XDECREF(parts[k]);
      Py_DECREF(a);
      if (bad) {
        if (!PyErr_Occurred())
          PyErr_Format(PyExc_ValueError, "molecule %zd: StereoChecks needs kind (n,) int32, idx (n, 5) int32, par (n, 2) float64", m);
        goto fail;
      }
      continue;
    }
    PyObject* checks = PySequence_Fast(a, "FlatMolecule.checks must be a sequence of (kind, idx, par)");
    Py_DECREF(a);
    if (checks == NULL) goto fail;
    const Py_ssize_t nc = PySequence_Fast_GET_SIZE(checks);
    if (used + nc > check_capacity) {
      Py_DECREF(checks);
      PyErr_SetString(PyExc_ValueError, "stereo-check arrays too small for the molecules' checks");
      goto fail;
    }
    d->n_checks   = (int32_t)nc;
    d->check_kind = check_kind + used;
    d->check_idx  = check_idx + 5 * used;
    d->check_par  = check_par + 2 * used;
    for (Py_ssize_t c = 0; c < nc; ++c, ++used) {
      PyObject* item = PySequence_Fast(PySequence_Fast_GET_ITEM(checks, c), "a stereo check is (kind, idx, par)");
      if (item == NULL || PySequence_Fast_GET_SIZE(item) != 3) {
        Py_XDECREF(item);
        Py_DECREF(checks);
        if (!PyErr_Occurred()) PyErr_Format(PyExc_ValueError, "molecule %zd: a stereo check is (kind, idx, par)", m);
        goto fail;
      }
      check_kind[used] = (int32_t)PyLong_AsLong(PySequence_Fast_GET_ITEM(item, 0));
      PyObject* idx    = PySequence_Fast(PySequence_Fast_GET_ITEM(item, 1), "check idx must be a sequence");
      PyObject* par    = idx ? PySequence_Fast(PySequence_Fast_GET_ITEM(item, 2), "check par must be a sequence") : NULL;
      if (idx == NULL || par == NULL || PySequence_Fast_GET_SIZE(idx) > 5 || PySequence_Fast_GET_SIZE(par) > 2) {
        Py_XDECREF(idx);
        Py_XDECREF(par);
        Py_DECREF(item);
        Py_DECREF(checks);
        if (!PyErr_Occurred()) PyErr_Format(PyExc_ValueError, "molecule %zd: a stereo check has at most 5 indices and 2 parameters", m);
        goto fail;
      }
      for (Py_ssize_t k = 0; k < 5; ++k)
        check_idx[5 * used + k] = k < PySequence_Fast_GET_SIZE(idx) ? (int32_t)PyLong_AsLong(PySequence_Fast_GET_ITEM(idx, k)) : 0;
      for (Py_ssize_t k = 0; k < 2; ++k)
        check_par[2 * used + k] = k < PySequence_Fast_GET_SIZE(par) ? PyFloat_AsDouble(PySequence_Fast_GET_ITEM(par, k)) : 0.0;
      Py_DECREF(idx);
      Py_DECREF(par);
      Py_DECREF(item);
      if (PyErr_Occurred()) {
        Py_DECREF(checks);
        goto fail;
      }
    }
    Py_DECREF(checks);
  }
  Py_DECREF(fast);
  return used;
fail:
  Py_DECREF(fast);
  return -1;
}

/* tables[m] = n_groups (idx, par) pairs -> out[m * n_groups + g].  Returns 0, -1 with a Python exception set. */
int nvmk_py_gather_term_tables(PyObject* tables, const int32_t* n_idx, const int32_t* n_par, int n_groups, nvmk_host_terms* out,
                               PyObject* keep, PyObject* convert) {
  PyObject* fast = PySequence_Fast(tables, "tables must be a sequence of per-molecule term groups");
  if (fast == NULL) return -1;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
  for (Py_ssize_t m = 0; m < n; ++m) {
    if (groups_of(PySequence_Fast_GET_ITEM(fast, m), n_idx, n_par, n_groups, keep, convert, out + m * n_groups, "term tables", m) != 0) {
      Py_DECREF(fast);
      return -1;
    }
  }
  Py_DECREF(fast);
  return 0;
}
