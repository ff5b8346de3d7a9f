/* ref_ply_dump.c -- ORACLE / TEST INFRASTRUCTURE ONLY (never linked into or called by the product).
 *
 * A driver of the reference's own vendored PLY parser, rply 1.01 (/root/reference/src/mesh/rply-1.01/rply.c, plain C, no
 * dependencies), used the way the reference's loader uses it (src/mesh/MeshBase.cpp:254-340 callbacks, :1443-1520 two passes):
 * count pass with ply_set_read_cb("vertex","x"|"nx"|"s"|"u") and ("face","vertex_indices"), then a data pass whose callbacks
 * write x y z / nx ny nz / s t u v at running cursors and the first three entries of each face list at a running triangle
 * cursor.  It pins fermat_amd's PLY reader (csrc/host/scene_io.cpp load_ply, scene.py load_ply) to the reference's behaviour.
 *
 * Built by oracle/Makefile into oracle/_ref/ply_dump from the rply source WHERE IT LIES under /root/reference (nothing is
 * copied); oracle/_ref/ is git-ignored.  usage: ply_dump in.ply out.bin
 * out.bin: int32 {nv, nn, nt, ntri}, float32 P[nv*3], N[nn*3], T[nt*2], int32 tri[ntri*3]            exit code 1 on any error
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "rply.h"

typedef struct
{
	float *P, *N, *T;
	int *tri;
	int nv, nn, nt, ntri;
	int cv, cn, ct, ctri;
	int bad;
} Sink;

static int on_vertex(p_ply_argument a)
{
	Sink* s; int coord;
	ply_get_argument_user_data(a, (void**)&s, &coord);
	const float value = (float)ply_get_argument_value(a);
	if (coord <= 2)      { if (s->cv >= s->nv) { s->bad = 1; return 0; } s->P[3 * s->cv + coord] = value;       if (coord == 2) s->cv++; }
	else if (coord <= 5) { if (s->cn >= s->nn) { s->bad = 1; return 0; } s->N[3 * s->cn + (coord - 3)] = value; if (coord == 5) s->cn++; }
	else                 { if (s->ct >= s->nt) { s->bad = 1; return 0; } s->T[2 * s->ct + (coord - 6)] = value; if (coord == 7) s->ct++; }
	return 1;
}

static int on_face(p_ply_argument a)
{
	Sink* s; int n, which;
	ply_get_argument_user_data(a, (void**)&s, NULL);
	ply_get_argument_property(a, NULL, &n, &which);
	if (which < 0 || which > 2) return 1;            /* the list length (-1) and the tail of longer faces are ignored */
	if (s->ctri >= s->ntri) { s->bad = 1; return 0; }
	s->tri[3 * s->ctri + which] = (int)ply_get_argument_value(a);
	if (which == 2) s->ctri++;
	return 1;
}

static void quiet(const char* m) { (void)m; }

int main(int argc, char** argv)
{
	if (argc != 3) { fprintf(stderr, "usage: ply_dump in.ply out.bin\n"); return 2; }
	Sink s; memset(&s, 0, sizeof(s));
	p_ply ply = ply_open(argv[1], quiet);
	if (!ply) return 1;
	if (!ply_read_header(ply)) { ply_close(ply); return 1; }
	s.nv = (int)ply_set_read_cb(ply, "vertex", "x", NULL, NULL, 0);
	s.nn = (int)ply_set_read_cb(ply, "vertex", "nx", NULL, NULL, 3);
	s.nt = (int)ply_set_read_cb(ply, "vertex", "s", NULL, NULL, 6);
	if (s.nt == 0) s.nt = (int)ply_set_read_cb(ply, "vertex", "u", NULL, NULL, 6);
	s.ntri = (int)ply_set_read_cb(ply, "face", "vertex_indices", NULL, NULL, 0);
	ply_close(ply);
	if (s.nv < 0 || s.ntri < 0) return 1;
	s.P = (float*)calloc((size_t)s.nv * 3 + 1, sizeof(float)); s.N = (float*)calloc((size_t)s.nn * 3 + 1, sizeof(float));
	s.T = (float*)calloc((size_t)s.nt * 2 + 1, sizeof(float)); s.tri = (int*)calloc((size_t)s.ntri * 3 + 1, sizeof(int));

	ply = ply_open(argv[1], quiet);
	if (!ply || !ply_read_header(ply)) return 1;
	ply_set_read_cb(ply, "vertex", "x", on_vertex, &s, 0);
	ply_set_read_cb(ply, "vertex", "y", on_vertex, &s, 1);
	ply_set_read_cb(ply, "vertex", "z", on_vertex, &s, 2);
	if (s.nn)
	{
		ply_set_read_cb(ply, "vertex", "nx", on_vertex, &s, 3);
		ply_set_read_cb(ply, "vertex", "ny", on_vertex, &s, 4);
		ply_set_read_cb(ply, "vertex", "nz", on_vertex, &s, 5);
	}
	if (s.nt)
	{
		ply_set_read_cb(ply, "vertex", "s", on_vertex, &s, 6);
		ply_set_read_cb(ply, "vertex", "t", on_vertex, &s, 7);
		ply_set_read_cb(ply, "vertex", "u", on_vertex, &s, 6);
		ply_set_read_cb(ply, "vertex", "v", on_vertex, &s, 7);
	}
	ply_set_read_cb(ply, "face", "vertex_indices", on_face, &s, 0);
	const int ok = ply_read(ply);
	ply_close(ply);
	if (!ok || s.bad) return 1;

	FILE* f = fopen(argv[2], "wb");
	if (!f) return 2;
	const int head[4] = { s.nv, s.nn, s.nt, s.ntri };
	fwrite(head, sizeof(int), 4, f);
	fwrite(s.P, sizeof(float), (size_t)s.nv * 3, f); fwrite(s.N, sizeof(float), (size_t)s.nn * 3, f);
	fwrite(s.T, sizeof(float), (size_t)s.nt * 2, f); fwrite(s.tri, sizeof(int), (size_t)s.ntri * 3, f);
	fclose(f);
	return 0;
}
