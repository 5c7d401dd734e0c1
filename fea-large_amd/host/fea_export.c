/*
 * fea_export.c -- Gmsh export and the per-load-step snapshots it needs.
 * Mirrors solver_export_tetrahedra10_gmsh (fea_solver.c:1375-1488) and
 * solver_load_step_init (:605-636) so that post-processing written for the
 * reference's .msh files (utilities/gmshanalyser.py) reads these unchanged.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fea_host.h"

void fea_export_name(const char *deck_path, char *out)
{
  const char *dot = strrchr(deck_path, '.');
  const char *slash = strrchr(deck_path, '/');
  size_t n = (dot && (!slash || dot > slash)) ? (size_t)(dot - deck_path) : strlen(deck_path);
  memcpy(out, deck_path, n);
  strcpy(out + n, ".msh");
}

void fea_snapshots_free(fea_step_snapshot *steps, int n)
{
  int i;
  if (!steps) return;
  for (i = 0; i < n; ++i) { free(steps[i].nodes); free(steps[i].stress0); steps[i].nodes = steps[i].stress0 = NULL; }
}

/* what solver_load_step_init keeps of a finished increment (fea_solver.c:605-636): the nodes and, for the
 * export, the stress of Gauss point 0 of every element (:1480) */
struct snap_sink { fea_step_snapshot *steps; int cap; double *S; };

static int keep_snapshot(const fea_deck *d, feahip_ctx *ctx, int step, void *user)
{
  struct snap_sink *k = (struct snap_sink *)user;
  const int G = d->gauss_nodes_count;
  int e, rc;
  if (!k->steps || step >= k->cap) return 0;
  k->steps[step].nodes = (double *)malloc(sizeof(double) * 3 * (size_t)d->nodes_count);
  k->steps[step].stress0 = (double *)malloc(sizeof(double) * 9 * (size_t)d->elements_count);
  if (!k->S) k->S = (double *)malloc(sizeof(double) * 9 * (size_t)d->elements_count * G);
  if (!k->steps[step].nodes || !k->steps[step].stress0 || !k->S) return FEAHIP_ENOMEM;
  if ((rc = feahip_get_nodes(ctx, k->steps[step].nodes))) return rc;
  if ((rc = feahip_get_stresses(ctx, k->S))) return rc;
  for (e = 0; e < d->elements_count; ++e)
    memcpy(k->steps[step].stress0 + (size_t)e * 9, k->S + (size_t)e * G * 9, sizeof(double) * 9);
  return 0;
}

int fea_solve_with_snapshots(const fea_deck *d, feahip_ctx *ctx, void *logp, fea_step_snapshot *steps, int cap)
{
  struct snap_sink k;
  int done;
  k.steps = steps; k.cap = cap; k.S = NULL;
  done = fea_solve_steps(d, ctx, logp, keep_snapshot, &k);     /* the one Newton loop of the host side (fea_solve.c) */
  free(k.S);
  return done;
}

int fea_export_gmsh(const char *filename, const fea_deck *d, const fea_step_snapshot *steps, int nsteps)
{
  FILE *f = fopen(filename, "w+");
  int i, j, k, load;
  const int npe = d->nodes_per_element;
  if (!f) return -1;
  fprintf(f, "$MeshFormat\n2.0 0 8\n$EndMeshFormat\n");
  fprintf(f, "$Nodes\n%d\n", d->nodes_count);
  for (i = 0; i < d->nodes_count; ++i)
    fprintf(f, "%d %f %f %f\n", i + 1, d->nodes[3 * i], d->nodes[3 * i + 1], d->nodes[3 * i + 2]);
  fprintf(f, "$EndNodes\n$Elements\n%d\n", d->elements_count);
  for (i = 0; i < d->elements_count; ++i) {
    const int *c = d->elements + (size_t)i * npe;
    if (npe == 10) {                       /* our 8 <-> Gmsh 9 (fea_solver.c:1430-1434) */
      fprintf(f, "%d 11 3 1 1 1 ", i + 1);
      for (j = 0; j < 8; ++j) fprintf(f, "%d ", c[j] + 1);
      fprintf(f, "%d %d \n", c[9] + 1, c[8] + 1);
    } else if (npe == 8) {                 /* Gmsh type 5, 8-node hexahedron: the same corner order */
      fprintf(f, "%d 5 3 1 1 1 ", i + 1);
      for (j = 0; j < 8; ++j) fprintf(f, "%d ", c[j] + 1);
      fprintf(f, "\n");
    } else {
      fprintf(f, "%d 4 3 1 1 1 ", i + 1);
      for (j = 0; j < 4; ++j) fprintf(f, "%d ", c[j] + 1);
      fprintf(f, "\n");
    }
  }
  fprintf(f, "$EndElements\n");
  for (load = 0; load <= nsteps; ++load) {                       /* :1440, load 0 = zeros */
    fprintf(f, "$NodeData\n1\n\"Displacements\"\n1\n%f\n3\n%d\n3\n%d\n", load * 0.83333333, load, d->nodes_count);
    for (i = 0; i < d->nodes_count; ++i) {
      double u[3] = {0, 0, 0};
      if (load)
        for (j = 0; j < 3; ++j) u[j] = steps[load - 1].nodes[3 * i + j] - d->nodes[3 * i + j];
      fprintf(f, "%d %f %f %f\n", i + 1, u[0], u[1], u[2]);
    }
    fprintf(f, "$EndNodeData\n");
    fprintf(f, "$ElementData\n1\n\"Stress tensor\"\n1\n%f\n3\n%d\n9\n%d\n", load * 0.83333333, load, d->elements_count);
    for (i = 0; i < d->elements_count; ++i) {
      fprintf(f, "%d ", i + 1);
      for (j = 0; j < 3; ++j)
        for (k = 0; k < 3; ++k)
          fprintf(f, "%f ", load ? steps[load - 1].stress0[(size_t)i * 9 + 3 * j + k] : 0.0);
      fprintf(f, "\n");
    }
    fprintf(f, "$EndElementData\n");
  }
  fclose(f);
  return 0;
}
