/*
 * sexp_reader.c -- streaming reader / writer for the task deck grammar.
 *
 * The reference parses the deck into an AST with libsexp and walks it
 * (solver-large/sexp_loader.c:249-327).  libsexp is not part of the
 * reference tree, and the 10M-element decks this build targets are GB-sized
 * as text, so this reader consumes the token stream directly and fills flat
 * arrays; nothing is kept but the current list's attributes.
 *
 * Grammar (from the five decks under solver-large/data and the emitter
 * utilities/tetgenProcessor/FEATask.hs:177-207):
 *   ';' starts a comment to end of line; lists are ( head item* );
 *   ':key value' pairs are attributes of the enclosing list; symbols are
 *   compared without regard to case (`yes` vs "YES", sexp_loader.c:90);
 *   numbers go through strtod, so coordinates round exactly as written.
 */
#include <ctype.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fea_host.h"

#define TOK_MAX 256
#define ATTR_MAX 16

typedef struct {
  FILE *f;
  int line;
  char err[256];
} reader;

enum { T_EOF, T_OPEN, T_CLOSE, T_ATOM };

static int next_token(reader *r, char *buf)
{
  int c;
  for (;;) {
    c = fgetc(r->f);
    if (c == EOF) return T_EOF;
    if (c == '\n') { r->line++; continue; }
    if (isspace(c)) continue;
    if (c == ';') {
      while ((c = fgetc(r->f)) != EOF && c != '\n') {}
      if (c == '\n') r->line++;
      continue;
    }
    break;
  }
  if (c == '(') return T_OPEN;
  if (c == ')') return T_CLOSE;
  {
    int n = 0;
    if (c == '"') {
      while ((c = fgetc(r->f)) != EOF && c != '"')
        if (n < TOK_MAX - 1) buf[n++] = (char)c;
    } else {
      do {
        if (n < TOK_MAX - 1) buf[n++] = (char)c;
        c = fgetc(r->f);
      } while (c != EOF && !isspace(c) && c != '(' && c != ')' && c != ';');
      if (c != EOF) ungetc(c, r->f);
    }
    buf[n] = 0;
  }
  return T_ATOM;
}

static int ieq(const char *a, const char *b)
{
  for (; *a && *b; ++a, ++b)
    if (toupper((unsigned char)*a) != toupper((unsigned char)*b)) return 0;
  return *a == 0 && *b == 0;
}

typedef struct { char key[48]; char val[TOK_MAX]; } attr;

static const char *attr_get(const attr *a, int n, const char *key)
{
  int i;
  for (i = 0; i < n; ++i)
    if (ieq(a[i].key, key)) return a[i].val;
  return NULL;
}

static int fail(reader *r, const char *msg)
{
  snprintf(r->err, sizeof(r->err), "line %d: %s", r->line, msg);
  return -1;
}

static int need_num(reader *r, const attr *a, int n, const char *key, double *out)
{
  const char *v = attr_get(a, n, key);
  char *end;
  if (!v) { char m[96]; snprintf(m, sizeof m, "missing attribute :%s", key); return fail(r, m); }
  *out = strtod(v, &end);
  if (end == v) { char m[96]; snprintf(m, sizeof m, "attribute :%s is not a number", key); return fail(r, m); }
  return 0;
}

/* skips a list whose '(' has been consumed */
static int skip_list(reader *r)
{
  char buf[TOK_MAX];
  int depth = 1, t;
  while (depth > 0) {
    t = next_token(r, buf);
    if (t == T_EOF) return fail(r, "unexpected end of file");
    if (t == T_OPEN) depth++;
    else if (t == T_CLOSE) depth--;
  }
  return 0;
}

/* (nodes (x y z) ...)  sexp_loader.c:169-189 */
static int read_nodes(reader *r, fea_deck *d)
{
  char buf[TOK_MAX];
  size_t cap = 1024, n = 0;
  double *p = (double *)malloc(cap * 3 * sizeof(double));
  for (;;) {
    int t = next_token(r, buf), k;
    if (t == T_CLOSE) break;
    if (t != T_OPEN) { free(p); return fail(r, "node entry must be a list of three numbers"); }
    if (n == cap) { cap *= 2; p = (double *)realloc(p, cap * 3 * sizeof(double)); }
    for (k = 0; k < 3; ++k) {
      char *end;
      if (next_token(r, buf) != T_ATOM) { free(p); return fail(r, "node needs three coordinates"); }
      p[n * 3 + k] = strtod(buf, &end);
      if (end == buf) { free(p); return fail(r, "bad node coordinate"); }
    }
    if (next_token(r, buf) != T_CLOSE) { free(p); return fail(r, "node needs exactly three coordinates"); }
    n++;
  }
  free(d->nodes);
  d->nodes = p; d->nodes_count = (int)n;
  return 0;
}

/* (elements (n0 ... n_{npe-1}) ...)  sexp_loader.c:191-211; the element
 * type must already be known, as in the reference (:203-206)                */
static int read_elements(reader *r, fea_deck *d)
{
  char buf[TOK_MAX];
  int npe = d->nodes_per_element;
  size_t cap = 1024, n = 0;
  int *p = (int *)malloc(cap * npe * sizeof(int));
  for (;;) {
    int t = next_token(r, buf), k;
    if (t == T_CLOSE) break;
    if (t != T_OPEN) { free(p); return fail(r, "element entry must be a list of node ids"); }
    if (n == cap) { cap *= 2; p = (int *)realloc(p, cap * npe * sizeof(int)); }
    for (k = 0; k < npe; ++k) {
      char *end;
      if (next_token(r, buf) != T_ATOM) { free(p); return fail(r, "element has too few node ids"); }
      p[n * npe + k] = (int)strtol(buf, &end, 10);
      if (end == buf) { free(p); return fail(r, "bad node id in element"); }
    }
    if (next_token(r, buf) != T_CLOSE) { free(p); return fail(r, "element has too many node ids"); }
    n++;
  }
  free(d->elements);
  d->elements = p; d->elements_count = (int)n;
  return 0;
}

/* reads the attributes of a list whose head has been consumed, up to ')';
 * nested lists are handed to `nested` (may be NULL = skip)                  */
typedef int (*nested_fn)(reader *r, fea_deck *d);
static int read_list(reader *r, fea_deck *d);

static int read_attrs(reader *r, fea_deck *d, attr *a, int *na, int recurse)
{
  char buf[TOK_MAX];
  *na = 0;
  for (;;) {
    int t = next_token(r, buf);
    if (t == T_EOF) return fail(r, "unexpected end of file");
    if (t == T_CLOSE) return 0;
    if (t == T_OPEN) {
      if (recurse) { if (read_list(r, d)) return -1; }
      else if (skip_list(r)) return -1;
      continue;
    }
    if (buf[0] == ':') {
      char key[48];
      snprintf(key, sizeof key, "%s", buf + 1);
      t = next_token(r, buf);
      if (t == T_OPEN) { if (skip_list(r)) return -1; continue; }
      if (t != T_ATOM) return fail(r, "attribute without a value");
      if (*na < ATTR_MAX) {
        snprintf(a[*na].key, sizeof a[*na].key, "%s", key);
        snprintf(a[*na].val, sizeof a[*na].val, "%s", buf);
        (*na)++;
      }
    }
  }
}

/* (prescribed-displacements (presc-node :x :y :z :type :node-id) ...)
 * sexp_loader.c:213-246 */
static int read_prescribed(reader *r, fea_deck *d)
{
  char buf[TOK_MAX];
  size_t cap = 256, n = 0;
  int *node = (int *)malloc(cap * sizeof(int)), *type = (int *)malloc(cap * sizeof(int));
  double *val = (double *)malloc(cap * 3 * sizeof(double));
  int rc = 0;
  for (;;) {
    attr a[ATTR_MAX];
    int na, t = next_token(r, buf);
    double v;
    if (t == T_CLOSE) break;
    if (t != T_OPEN || next_token(r, buf) != T_ATOM || !ieq(buf, "presc-node")) {
      rc = fail(r, "expected (presc-node ...)"); break;
    }
    if ((rc = read_attrs(r, d, a, &na, 0))) break;
    if (n == cap) {
      cap *= 2;
      node = (int *)realloc(node, cap * sizeof(int));
      type = (int *)realloc(type, cap * sizeof(int));
      val = (double *)realloc(val, cap * 3 * sizeof(double));
    }
    if ((rc = need_num(r, a, na, "node-id", &v))) break; node[n] = (int)v;
    if ((rc = need_num(r, a, na, "x", &val[n * 3 + 0]))) break;
    if ((rc = need_num(r, a, na, "y", &val[n * 3 + 1]))) break;
    if ((rc = need_num(r, a, na, "z", &val[n * 3 + 2]))) break;
    if ((rc = need_num(r, a, na, "type", &v))) break; type[n] = (int)v;
    n++;
  }
  if (rc) { free(node); free(type); free(val); return rc; }
  free(d->presc_node); free(d->presc_type); free(d->presc_values);
  d->presc_node = node; d->presc_type = type; d->presc_values = val;
  d->prescribed_nodes_count = (int)n;
  return 0;
}

/* one list, '(' consumed: dispatch on the head like traverse_function
 * (sexp_loader.c:249-272) */
static int read_list(reader *r, fea_deck *d)
{
  char head[TOK_MAX];
  attr a[ATTR_MAX];
  int na, t = next_token(r, head);
  double v;
  const char *s;
  if (t == T_CLOSE) return 0;
  if (t == T_OPEN) {                     /* list of lists */
    if (read_list(r, d)) return -1;
    head[0] = 0;
  } else if (t == T_EOF)
    return fail(r, "unexpected end of file");

  if (ieq(head, "nodes")) return read_nodes(r, d);
  if (ieq(head, "elements")) return read_elements(r, d);
  if (ieq(head, "prescribed-displacements")) return read_prescribed(r, d);

  if (read_attrs(r, d, a, &na, 1)) return -1;

  if (ieq(head, "model")) {                                   /* :32-54 */
    if ((s = attr_get(a, na, "name"))) {
      if (ieq(s, "A5")) { d->model = FEAHIP_MODEL_A5; d->parameters_count = 2; }
      else if (ieq(s, "COMPRESSIBLE_NEOHOOKEAN")) { d->model = FEAHIP_MODEL_COMPRESSIBLE_NEOHOOKEAN; d->parameters_count = 2; }
      else { char m[TOK_MAX + 32]; snprintf(m, sizeof m, "unknown model type '%s'", s); return fail(r, m); }
    }
  } else if (ieq(head, "model-parameters")) {                 /* :56-73 */
    if (need_num(r, a, na, "lambda", &d->parameters[0])) return -1;
    if (need_num(r, a, na, "mu", &d->parameters[1])) return -1;
  } else if (ieq(head, "solution")) {                         /* :75-95 */
    if (need_num(r, a, na, "desired-tolerance", &d->desired_tolerance)) return -1;
    if (!attr_get(a, na, "task-type")) return fail(r, "missing attribute :task-type");
    if (need_num(r, a, na, "load-increments-count", &v)) return -1;
    d->load_increments_count = (int)v;
    if (!(s = attr_get(a, na, "modified-newton"))) return fail(r, "missing attribute :modified-newton");
    d->modified_newton = (ieq(s, "YES") || ieq(s, "TRUE")) ? 1 : 0;
    if (need_num(r, a, na, "max-newton-count", &v)) return -1;
    d->max_newton_count = (int)v;
  } else if (ieq(head, "slae-solver")) {                      /* :97-138 */
    d->solver_type = FEAHIP_CG; d->solver_tolerance = 1e-14; d->solver_max_iter = 20000;
    if ((s = attr_get(a, na, "type"))) {
      if (ieq(s, "CG") || ieq(s, "PCG_ILU")) {
        d->solver_type = ieq(s, "CG") ? FEAHIP_CG : FEAHIP_PCG_ILU;
        if (attr_get(a, na, "tolerance") && need_num(r, a, na, "tolerance", &d->solver_tolerance)) return -1;
        if (attr_get(a, na, "max-iterations")) {
          if (need_num(r, a, na, "max-iterations", &v)) return -1;
          d->solver_max_iter = (int)v;
        }
      } else if (ieq(s, "CHOLESKY")) d->solver_type = FEAHIP_CHOLESKY;
      else { char m[TOK_MAX + 32]; snprintf(m, sizeof m, "unknown solver type '%s'", s); return fail(r, m); }
    }
  } else if (ieq(head, "element-type")) {                     /* :141-155 */
    if (need_num(r, a, na, "gauss-nodes-count", &v)) return -1;
    d->gauss_nodes_count = (int)v;
    if (need_num(r, a, na, "nodes-count", &v)) return -1;
    d->nodes_per_element = (int)v;
    if (!(s = attr_get(a, na, "name"))) return fail(r, "missing attribute :name");
    if (ieq(s, "TETRAHEDRA10")) d->ele_type = FEA_TETRAHEDRA10;
    else if (ieq(s, "TETRAHEDRA4")) d->ele_type = FEA_TETRAHEDRA4;   /* build extension */
    else if (ieq(s, "HEXAHEDRA8")) d->ele_type = FEA_HEXAHEDRA8;     /* build extension */
    else { char m[TOK_MAX + 32]; snprintf(m, sizeof m, "unknown element type '%s'", s); return fail(r, m); }
  } else if (ieq(head, "line-search")) {                      /* :157-163 */
    if (need_num(r, a, na, "max", &v)) return -1;
    d->linesearch_max = (int)v;
  } else if (ieq(head, "arc-length")) {                       /* :165-171 */
    if (need_num(r, a, na, "max", &v)) return -1;
    d->arclength_max = (int)v;
  }
  return 0;
}

static void deck_defaults(fea_deck *d)
{
  memset(d, 0, sizeof(*d));
  d->desired_tolerance = 1e-8;                 /* fea_solver.c:1514-1527 */
  d->ele_type = FEA_TETRAHEDRA10;
  d->modified_newton = 1;
  d->model = FEAHIP_MODEL_A5;
  d->parameters_count = 2;
  d->parameters[0] = 100; d->parameters[1] = 100;
  d->gauss_nodes_count = 5;                    /* :1546-1547 */
  d->nodes_per_element = 10;
  d->solver_type = FEAHIP_CG;                  /* sexp_loader.c:101-103 */
  d->solver_tolerance = 1e-14;
  d->solver_max_iter = 20000;
}

int fea_deck_load(const char *path, fea_deck *deck, char *errbuf, int errlen)
{
  reader r;
  char buf[TOK_MAX];
  int t, rc;
  deck_defaults(deck);
  r.line = 1; r.err[0] = 0;
  r.f = fopen(path, "rt");
  if (!r.f) {
    if (errbuf) snprintf(errbuf, (size_t)errlen, "could not open file %s", path);
    return -1;
  }
  t = next_token(&r, buf);
  if (t != T_OPEN || next_token(&r, buf) != T_ATOM || !ieq(buf, "task")) {
    fclose(r.f);
    if (errbuf) snprintf(errbuf, (size_t)errlen, "deck does not start with (task");
    return -1;
  }
  {
    attr a[ATTR_MAX];
    int na;
    rc = read_attrs(&r, deck, a, &na, 1);
  }
  fclose(r.f);
  if (rc == 0) {
    int i;
    if (deck->nodes_count == 0 || deck->elements_count == 0) { rc = -1; snprintf(r.err, sizeof r.err, "deck has no nodes or no elements"); }
    for (i = 0; rc == 0 && i < deck->elements_count * deck->nodes_per_element; ++i)
      if (deck->elements[i] < 0 || deck->elements[i] >= deck->nodes_count) {
        rc = -1; snprintf(r.err, sizeof r.err, "element %d refers to node %d outside the node list", i / deck->nodes_per_element, deck->elements[i]);
      }
  }
  if (rc) {
    if (errbuf) snprintf(errbuf, (size_t)errlen, "%s", r.err);
    fea_deck_free(deck);
    return -1;
  }
  return 0;
}

void fea_deck_free(fea_deck *d)
{
  if (!d) return;
  free(d->nodes); free(d->elements);
  free(d->presc_node); free(d->presc_type); free(d->presc_values);
  d->nodes = NULL; d->elements = NULL;
  d->presc_node = d->presc_type = NULL; d->presc_values = NULL;
  d->nodes_count = d->elements_count = d->prescribed_nodes_count = 0;
}

int fea_deck_save(const char *path, const fea_deck *d)
{
  static const char *solver[] = {"CG", "PCG_ILU", "CHOLESKY"};
  FILE *f = fopen(path, "wt");
  int i, k;
  if (!f) return -1;
  fprintf(f, ";; -*- Mode: lisp; -*-\n(task\n");
  fprintf(f, " (model :name %s\n        (model-parameters :mu %.17g :lambda %.17g))\n",
          d->model == FEAHIP_MODEL_A5 ? "A5" : "COMPRESSIBLE_NEOHOOKEAN", d->parameters[1], d->parameters[0]);
  fprintf(f, " (solution :desired-tolerance %.17g :task-type CARTESIAN3D :load-increments-count %d"
             " :modified-newton %s :max-newton-count %d\n",
          d->desired_tolerance, d->load_increments_count, d->modified_newton ? "yes" : "no", d->max_newton_count);
  fprintf(f, "   (element-type :gauss-nodes-count %d :name %s :nodes-count %d)\n", d->gauss_nodes_count,
          d->ele_type == FEA_TETRAHEDRA4 ? "TETRAHEDRA4" : d->ele_type == FEA_HEXAHEDRA8 ? "HEXAHEDRA8" : "TETRAHEDRA10", d->nodes_per_element);
  fprintf(f, "   (slae-solver :type %s :tolerance %.17g :max-iterations %d)\n", solver[d->solver_type],
          d->solver_tolerance, d->solver_max_iter);
  fprintf(f, "   (line-search :max %d)\n   (arc-length :max %d))\n", d->linesearch_max, d->arclength_max);
  fprintf(f, " (input-data\n  (geometry\n   (nodes");
  for (i = 0; i < d->nodes_count; ++i)
    fprintf(f, "\n    (%.17g %.17g %.17g)", d->nodes[3 * i], d->nodes[3 * i + 1], d->nodes[3 * i + 2]);
  fprintf(f, ")\n   (elements");
  for (i = 0; i < d->elements_count; ++i) {
    fprintf(f, "\n    (");
    for (k = 0; k < d->nodes_per_element; ++k)
      fprintf(f, k ? " %d" : "%d", d->elements[(size_t)i * d->nodes_per_element + k]);
    fprintf(f, ")");
  }
  fprintf(f, "))\n  (boundary-conditions\n   (prescribed-displacements");
  for (i = 0; i < d->prescribed_nodes_count; ++i)
    fprintf(f, "\n    (presc-node :y %.17g :x %.17g :z %.17g :type %d :node-id %d)", d->presc_values[3 * i + 1],
            d->presc_values[3 * i], d->presc_values[3 * i + 2], d->presc_type[i], d->presc_node[i]);
  fprintf(f, "))))\n");
  return fclose(f);
}
