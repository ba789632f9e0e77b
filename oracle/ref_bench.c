/* ref_bench.c -- repo-owned timing driver for the REFERENCE's extract path.
 *
 * TEST INFRASTRUCTURE ONLY (bench.py cpu_baseline, kind "reference").
 * Written against the public otezip/zip.h API alone; oracle/Makefile links it
 * with /root/reference/src/lib/otezip.c into oracle/_ref/ref_bench.
 *
 * Usage: ref_bench ARCHIVE PART NPARTS [MAX_ENTRIES]
 *   Opens its own zip_t, then zip_fopen_index()/zip_fclose() every entry with
 *   index % NPARTS == PART (independent handles per worker, as BASELINE.md
 *   section 4 prescribes: a zip_t is not shareable, ref src/lib/otezip.c:403-465).
 *   Prints one line: "entries bytes seconds null_entries crc_xor".
 */
#define _POSIX_C_SOURCE 200809L
#include <otezip/zip.h>

#include <stdio.h>
#include <stdlib.h>
#include <time.h>

static double now(void) {
	struct timespec ts;
	clock_gettime (CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char **argv) {
	if (argc < 4) {
		fprintf (stderr, "usage: %s archive part nparts [max_entries]\n", argv[0]);
		return 2;
	}
	int part = atoi (argv[2]);
	int nparts = atoi (argv[3]);
	long max_entries = argc > 4 ? atol (argv[4]) : -1;
	int err = 0;
	zip_t *za = zip_open (argv[1], ZIP_RDONLY, &err);
	if (!za) {
		fprintf (stderr, "zip_open failed: %d\n", err);
		return 1;
	}
	zip_uint64_t n = zip_get_num_files (za);
	if (max_entries >= 0 && (zip_uint64_t)max_entries < n) {
		n = (zip_uint64_t)max_entries;
	}
	unsigned long long bytes = 0, entries = 0, nulls = 0;
	unsigned int sum = 0;
	double t0 = now ();
	for (zip_uint64_t i = (zip_uint64_t)part; i < n; i += (zip_uint64_t)nparts) {
		zip_file_t *zf = zip_fopen_index (za, i, 0);
		if (!zf) {
			nulls++;
			continue;
		}
		bytes += zf->size;
		entries++;
		if (zf->size) {
			sum ^= (unsigned int)zf->data[0] | ((unsigned int)zf->data[zf->size - 1] << 8);
		}
		zip_fclose (zf);
	}
	double t1 = now ();
	zip_close (za);
	printf ("%llu %llu %.6f %llu %u\n", entries, bytes, t1 - t0, nulls, sum);
	return 0;
}
