/* zip_roundtrip.c -- drives the libzip-subset API the way the reference's own unit tests do
 * (test/unit/test_empty_zip.c, test/unit/test_set_file_compression.c, and the extract loop of
 * src/main.c:429-577), against libotezip_amd.so.
 *   zip_roundtrip list ARCHIVE            -> "n_entries\nname method comp uncomp crc\n..."
 *   zip_roundtrip extract ARCHIVE OUTDIR  -> writes every entry to OUTDIR/<index>, prints "i size|NULL"
 *   zip_roundtrip setcomp PATH            -> the test_set_file_compression scenario; exit 0 on success
 *   zip_roundtrip extract2 ARCHIVE OUTDIR VERIFY_CRC IGNORE_ZIPBOMB [rev]
 *                                         -> extract with the reference's global switches set; "rev": last entry first
 *   zip_roundtrip bench-extract ARCHIVE STRIDE
 *                                         -> times zip_open .. zip_fopen_index/zip_fclose of every entry .. zip_close;
 *                                            prints "i crc32 size" for every STRIDE-th entry, then "TOTAL bytes seconds"
 *   zip_roundtrip bench-add PATH INPUT N SIZE
 *                                         -> adds N buffers of SIZE bytes read from INPUT with default method STORE,
 *                                            sets DEFLATE on every one, closes; prints "TOTAL bytes seconds"
 *   zip_roundtrip add-list PATH INPUT SIZES [STAT_EVERY]
 *                                         -> ragged adds (sizes from the text file SIZES), DEFLATE set on each
 *   zip_roundtrip replace PATH            -> append + zip_file_replace + read-back scenario; exit 0 on success
 *   zip_roundtrip multi ARCHIVE K THREADS -> K handles on the same archive walked entry by entry, interleaved on one
 *                                            thread (THREADS = 0) or one thread per handle; prints "TOTAL bytes bad"
 *   zip_roundtrip add-fail PATH N         -> N adds whose flush is expected to FAIL (no usable device): zip_close must
 *                                            return -1 and leave an archive that lists only what it holds
 */
#define _POSIX_C_SOURCE 200809L
#include <otezip/zip.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <pthread.h>

static double now_s(void) {
	struct timespec ts;
	clock_gettime (CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* CRC-32 of the caller's view of an entry (the test compares it with zlib's) */
static uint32_t crc32_buf(const uint8_t *p, size_t n) {
	static uint32_t tab[256];
	if (!tab[1]) {
		for (uint32_t i = 0; i < 256; i++) {
			uint32_t c = i;
			for (int k = 0; k < 8; k++) {
				c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
			}
			tab[i] = c;
		}
	}
	uint32_t c = 0xFFFFFFFFu;
	for (size_t i = 0; i < n; i++) {
		c = tab[(c ^ p[i]) & 0xFF] ^ (c >> 8);
	}
	return ~c;
}

static int do_list(const char *path) {
	int err = -1;
	zip_t *za = zip_open (path, ZIP_RDONLY, &err);
	if (!za) {
		printf ("open-failed %d\n", err);
		return 1;
	}
	printf ("%llu\n", (unsigned long long)zip_get_num_files (za));
	for (zip_uint64_t i = 0; i < zip_get_num_files (za); i++) {
		zip_stat_t st;
		if (zip_stat_index (za, i, 0, &st) != 0) {
			return 2;
		}
		if (zip_name_locate (za, st.name, 0) < 0 || strcmp (zip_get_name (za, i, 0), st.name)) {
			return 3;
		}
		printf ("%s %u %llu %llu %08x\n", st.name, st.comp_method, (unsigned long long)st.comp_size,
			(unsigned long long)st.size, st.crc);
	}
	return zip_close (za);
}

static int do_extract(const char *path, const char *outdir) {
	int err = -1;
	zip_t *za = zip_open (path, ZIP_RDONLY, &err);
	if (!za) {
		printf ("open-failed %d\n", err);
		return 1;
	}
	for (zip_uint64_t i = 0; i < zip_get_num_files (za); i++) {
		zip_file_t *zf = zip_fopen_index (za, i, 0);
		if (!zf) {
			printf ("%llu NULL\n", (unsigned long long)i);
			continue;
		}
		char name[4096];
		snprintf (name, sizeof (name), "%s/%llu", outdir, (unsigned long long)i);
		FILE *f = fopen (name, "wb");
		/* half through zf->data (what callers of the reference do), half through zip_fread */
		uint32_t half = zf->size / 2;
		fwrite (zf->data, 1, half, f);
		zf->pos = half;
		char buf[777];
		zip_int64_t n;
		while ((n = zip_fread (zf, buf, sizeof (buf))) > 0) {
			fwrite (buf, 1, (size_t)n, f);
		}
		fclose (f);
		printf ("%llu %u\n", (unsigned long long)i, zf->size);
		zip_fclose (zf);
	}
	return zip_close (za);
}

static int do_setcomp(const char *path) {
	enum { N = 4096 };
	char expected[N];
	memset (expected, 'A', N);
	int err = -1;
	zip_t *za = zip_open (path, ZIP_CREATE | ZIP_TRUNCATE, &err);
	if (!za) {
		return 10;
	}
	za->default_method = ZIP_CM_DEFLATE; /* what the reference's CLI does for -z deflate (src/main.c:190) */
	char *buf = (char *)malloc (N);
	memcpy (buf, expected, N);
	zip_source_t *src = zip_source_buffer (za, buf, N, 1);
	zip_int64_t idx = zip_file_add (za, "hello.txt", src, 0);
	if (idx < 0) {
		return 11;
	}
	if (zip_set_file_compression (za, (zip_uint64_t)idx, ZIP_CM_DEFLATE, 0) != 0) {
		return 12;
	}
	zip_source_t *s2 = zip_source_buffer (za, "", 0, 0);
	if (zip_file_add (za, "empty", s2, 0) != 1) {
		return 13;
	}
	if (zip_close (za) != 0) {
		return 14;
	}
	za = zip_open (path, ZIP_RDONLY, &err);
	if (!za || zip_get_num_files (za) != 2) {
		return 15;
	}
	const char *name = zip_get_name (za, 0, 0);
	if (!name || strcmp (name, "hello.txt")) {
		return 16;
	}
	zip_file_t *zf = zip_fopen_index (za, 0, 0);
	if (!zf) {
		return 17; /* the reference fails here ("zip_fopen_index failed") */
	}
	if (zf->size != N || memcmp (zf->data, expected, N)) {
		return 18;
	}
	zip_fclose (zf);
	zf = zip_fopen_index (za, 1, 0);
	if (!zf || zf->size != 0) {
		return 19;
	}
	zip_fclose (zf);
	zip_close (za);
	return 0;
}

static int do_extract2(const char *path, const char *outdir, int verify, int ignore, int rev) {
	otezip_verify_crc = verify;
	otezip_ignore_zipbomb = ignore;
	int err = -1;
	zip_t *za = zip_open (path, ZIP_RDONLY, &err);
	if (!za) {
		printf ("open-failed %d\n", err);
		return 1;
	}
	const zip_uint64_t n = zip_get_num_files (za);
	for (zip_uint64_t j = 0; j < n; j++) {
		const zip_uint64_t i = rev ? n - 1 - j : j;
		zip_file_t *zf = zip_fopen_index (za, i, 0);
		if (!zf) {
			printf ("%llu NULL\n", (unsigned long long)i);
			continue;
		}
		char name[4096];
		snprintf (name, sizeof (name), "%s/%llu", outdir, (unsigned long long)i);
		FILE *f = fopen (name, "wb");
		fwrite (zf->data, 1, zf->size, f);
		fclose (f);
		printf ("%llu %u\n", (unsigned long long)i, zf->size);
		zip_fclose (zf);
	}
	return zip_close (za);
}

static int do_bench_extract(const char *path, unsigned stride) {
	const double t0 = now_s ();
	int err = -1;
	zip_t *za = zip_open (path, ZIP_RDONLY, &err);
	if (!za) {
		printf ("open-failed %d\n", err);
		return 1;
	}
	const zip_uint64_t n = zip_get_num_files (za);
	unsigned long long total = 0, touched = 0;
	int bad = 0;
	for (zip_uint64_t i = 0; i < n; i++) {
		zip_file_t *zf = zip_fopen_index (za, i, 0);
		if (!zf) {
			bad++;
			continue;
		}
		total += zf->size;
		touched += zf->size ? zf->data[0] + zf->data[zf->size - 1] : 0; /* the caller does look at the bytes */
		if (stride && i % stride == 0) {
			printf ("%llu %08x %u\n", (unsigned long long)i, crc32_buf (zf->data, zf->size), zf->size);
		}
		zip_fclose (zf);
	}
	zip_close (za);
	const double t1 = now_s ();
	printf ("TOTAL %llu %.6f %d %llu\n", total, t1 - t0, bad, touched);
	return bad ? 2 : 0;
}

static int do_bench_add(const char *path, const char *input, unsigned n, unsigned size, unsigned passes) {
	FILE *in = fopen (input, "rb");
	if (!in) {
		return 20;
	}
	uint8_t *all = (uint8_t *)malloc ((size_t)n * size);
	if (!all || fread (all, 1, (size_t)n * size, in) != (size_t)n * size) {
		return 21;
	}
	fclose (in);
	for (unsigned pass = 0; pass < passes; pass++) { /* (the first pass of a process creates the GPU context) */
		const double t0 = now_s ();
		int err = -1;
		zip_t *za = zip_open (path, ZIP_CREATE | ZIP_TRUNCATE, &err);
		if (!za) {
			return 22;
		}
		za->default_method = ZIP_CM_STORE;
		for (unsigned i = 0; i < n; i++) {
			char name[64];
			snprintf (name, sizeof (name), "f%05u", i);
			zip_source_t *src = zip_source_buffer (za, all + (size_t)i * size, size, 0);
			if (zip_file_add (za, name, src, 0) != (zip_int64_t)i) {
				return 23;
			}
		}
		for (unsigned i = 0; i < n; i++) {
			if (zip_set_file_compression (za, i, ZIP_CM_DEFLATE, 0) != 0) {
				return 24;
			}
		}
		if (zip_close (za) != 0) {
			return 25;
		}
		const double t1 = now_s ();
		printf ("TOTAL %llu %.6f\n", (unsigned long long)n * size, t1 - t0);
	}
	free (all);
	return 0;
}

/* K handles on one archive, every entry of each opened in turn */
struct walker {
	const char *path;
	zip_t *za;
	unsigned long long total;
	int bad;
};
static void walk_one(struct walker *w, zip_uint64_t i) {
	zip_file_t *zf = zip_fopen_index (w->za, i, 0);
	if (!zf) {
		w->bad++;
		return;
	}
	w->total += zf->size + (zf->size ? zf->data[zf->size - 1] : 0) * 0u;
	if (zf->size && crc32_buf (zf->data, zf->size) == 0) {
		w->bad += 0; /* (the bytes are read) */
	}
	zip_fclose (zf);
}
static void *walk_thread(void *p) {
	struct walker *w = (struct walker *)p;
	const zip_uint64_t n = zip_get_num_files (w->za);
	for (zip_uint64_t i = 0; i < n; i++) {
		walk_one (w, i);
	}
	return NULL;
}
static int do_multi(const char *path, int k, int threads) {
	struct walker w[16];
	if (k < 1 || k > 16) {
		return 64;
	}
	for (int a = 0; a < k; a++) {
		int err = -1;
		w[a].path = path;
		w[a].total = 0;
		w[a].bad = 0;
		w[a].za = zip_open (path, ZIP_RDONLY, &err);
		if (!w[a].za) {
			return 30;
		}
	}
	if (threads) {
		pthread_t th[16];
		for (int a = 0; a < k; a++) {
			if (pthread_create (&th[a], NULL, walk_thread, &w[a]) != 0) {
				return 31;
			}
		}
		for (int a = 0; a < k; a++) {
			pthread_join (th[a], NULL);
		}
	} else {
		const zip_uint64_t n = zip_get_num_files (w[0].za);
		for (zip_uint64_t i = 0; i < n; i++) {
			for (int a = 0; a < k; a++) {
				walk_one (&w[a], i);
			}
		}
	}
	unsigned long long total = 0;
	int bad = 0;
	for (int a = 0; a < k; a++) {
		total += w[a].total;
		bad += w[a].bad;
		zip_close (w[a].za);
	}
	printf ("TOTAL %llu %d\n", total, bad);
	return bad ? 2 : 0;
}

static int do_add_fail(const char *path, unsigned n) {
	int err = -1;
	zip_t *za = zip_open (path, ZIP_CREATE | ZIP_TRUNCATE, &err);
	if (!za) {
		return 40;
	}
	za->default_method = ZIP_CM_DEFLATE;
	static const char text[] = "the quick brown fox jumps over the lazy dog, again and again and again and again";
	for (unsigned i = 0; i < n; i++) {
		char name[64];
		snprintf (name, sizeof (name), "x%03u", i);
		zip_source_t *src = zip_source_buffer (za, text, sizeof (text), 0);
		if (!src) {
			return 41;
		}
		if (zip_file_add (za, name, src, 0) < 0) {
			zip_source_free (src); /* (a later add may already be refused: the failure is known) */
		}
	}
	const zip_uint64_t before = zip_get_num_files (za);
	const int rc = zip_close (za);
	printf ("CLOSE %d %llu\n", rc, (unsigned long long)before);
	return rc == -1 ? 0 : 42;
}

/* ragged adds: SIZES holds one decimal size per line; the buffers follow one another in INPUT.  Every
 * stat_every-th add is followed by a zip_stat_index of the new entry (which forces the pending batch out). */
static int do_add_list(const char *path, const char *input, const char *sizes, unsigned stat_every) {
	FILE *in = fopen (input, "rb"), *sf = fopen (sizes, "r");
	if (!in || !sf) {
		return 70;
	}
	int err = -1;
	zip_t *za = zip_open (path, ZIP_CREATE | ZIP_TRUNCATE, &err);
	if (!za) {
		return 71;
	}
	za->default_method = ZIP_CM_STORE;
	unsigned long sz;
	unsigned i = 0;
	while (fscanf (sf, "%lu", &sz) == 1) {
		uint8_t *buf = (uint8_t *)malloc (sz ? sz : 1);
		if (!buf || fread (buf, 1, sz, in) != sz) {
			return 72;
		}
		char name[64];
		snprintf (name, sizeof (name), "e%06u", i);
		zip_source_t *src = zip_source_buffer (za, buf, sz, 1); /* freep: the library owns buf */
		if (zip_file_add (za, name, src, 0) != (zip_int64_t)i) {
			return 73;
		}
		if (zip_set_file_compression (za, i, ZIP_CM_DEFLATE, 0) != 0) {
			return 74;
		}
		if (stat_every && (i + 1) % stat_every == 0) {
			zip_stat_t st;
			if (zip_stat_index (za, i, 0, &st) != 0 || st.size != sz) {
				return 75;
			}
		}
		i++;
	}
	fclose (in);
	fclose (sf);
	return zip_close (za) == 0 ? 0 : 76;
}

/* append mode, zip_file_replace and reading an entry back in the same session */
static int do_replace(const char *path) {
	int err = -1;
	zip_t *za = zip_open (path, ZIP_CREATE | ZIP_TRUNCATE, &err);
	if (!za) {
		return 30;
	}
	za->default_method = ZIP_CM_DEFLATE;
	static char a[3000], b[5000], c[100];
	memset (a, 'a', sizeof (a));
	memset (b, 'b', sizeof (b));
	memset (c, 'c', sizeof (c));
	if (zip_file_add (za, "a", zip_source_buffer (za, a, sizeof (a), 0), 0) != 0 ||
		zip_file_add (za, "b", zip_source_buffer (za, b, sizeof (b), 0), 0) != 1) {
		return 31;
	}
	zip_stat_t st;
	if (zip_stat_index (za, 0, 0, &st) != 0 || st.size != sizeof (a) || st.comp_size == 0 || st.comp_size >= sizeof (a)) {
		return 32; /* stat of an added entry reports what was really written */
	}
	if (zip_close (za) != 0) {
		return 33;
	}
	/* append */
	za = zip_open (path, ZIP_CREATE, &err);
	if (!za || zip_get_num_files (za) != 2) {
		return 34;
	}
	za->default_method = ZIP_CM_DEFLATE;
	if (zip_file_add (za, "c", zip_source_buffer (za, c, sizeof (c), 0), 0) != 2) {
		return 35;
	}
	/* read an OLD entry, replace it, read it again: the new bytes must come back */
	zip_file_t *zf = zip_fopen_index (za, 1, 0);
	if (!zf || zf->size != sizeof (b) || zf->data[0] != 'b') {
		return 36;
	}
	zip_fclose (zf);
	static char b2[7000];
	memset (b2, 'B', sizeof (b2));
	zip_source_t *rs = zip_source_buffer (za, b2, sizeof (b2), 0);
	if (zip_file_replace (za, 1, rs, 0) != 0) {
		return 37;
	}
	zip_source_free (rs); /* replace does not consume the source (ref :1617-1658) */
	zf = zip_fopen_index (za, 1, 0);
	if (!zf || zf->size != sizeof (b2) || zf->data[0] != 'B' || zf->data[sizeof (b2) - 1] != 'B') {
		return 38;
	}
	zip_fclose (zf);
	zf = zip_fopen_index (za, 2, 0); /* the entry added in this session */
	if (!zf || zf->size != sizeof (c) || zf->data[5] != 'c') {
		return 39;
	}
	zip_fclose (zf);
	if (zip_close (za) != 0) {
		return 40;
	}
	za = zip_open (path, ZIP_RDONLY, &err);
	if (!za || zip_get_num_files (za) != 3) {
		return 41;
	}
	const size_t want[3] = { sizeof (a), sizeof (b2), sizeof (c) };
	const char ch[3] = { 'a', 'B', 'c' };
	for (int i = 2; i >= 0; i--) { /* backwards: the replaced entry lies behind the others in the file */
		zf = zip_fopen_index (za, (zip_uint64_t)i, 0);
		if (!zf || zf->size != want[i] || zf->data[0] != (uint8_t)ch[i] || zf->data[want[i] - 1] != (uint8_t)ch[i]) {
			return 42 + i;
		}
		zip_fclose (zf);
	}
	return zip_close (za);
}

int main(int argc, char **argv) {
	if (argc >= 3 && !strcmp (argv[1], "list")) {
		return do_list (argv[2]);
	}
	if (argc >= 4 && !strcmp (argv[1], "extract")) {
		return do_extract (argv[2], argv[3]);
	}
	if (argc >= 3 && !strcmp (argv[1], "setcomp")) {
		return do_setcomp (argv[2]);
	}
	if (argc >= 6 && !strcmp (argv[1], "extract2")) {
		return do_extract2 (argv[2], argv[3], atoi (argv[4]), atoi (argv[5]), argc >= 7 && !strcmp (argv[6], "rev"));
	}
	if (argc >= 4 && !strcmp (argv[1], "bench-extract")) {
		/* optional 4th argument: repeat count (the first pass pays for creating the GPU context and for
		 * pinning the staging buffers; later passes of the same process do not) */
		int rc = 0;
		for (int k = 0; k < (argc >= 5 ? atoi (argv[4]) : 1) && rc == 0; k++) {
			rc = do_bench_extract (argv[2], (unsigned)atoi (argv[3]));
		}
		return rc;
	}
	if (argc >= 6 && !strcmp (argv[1], "bench-add")) {
		return do_bench_add (argv[2], argv[3], (unsigned)atoi (argv[4]), (unsigned)atoi (argv[5]), argc >= 7 ? (unsigned)atoi (argv[6]) : 1u);
	}
	if (argc >= 5 && !strcmp (argv[1], "add-list")) {
		return do_add_list (argv[2], argv[3], argv[4], argc >= 6 ? (unsigned)atoi (argv[5]) : 0);
	}
	if (argc >= 3 && !strcmp (argv[1], "replace")) {
		return do_replace (argv[2]);
	}
	if (argc >= 5 && !strcmp (argv[1], "multi")) {
		return do_multi (argv[2], atoi (argv[3]), atoi (argv[4]));
	}
	if (argc >= 4 && !strcmp (argv[1], "add-fail")) {
		return do_add_fail (argv[2], (unsigned)atoi (argv[3]));
	}
	return 64;
}
