/* zip_roundtrip.c -- drives the libzip-subset API the way the reference's own unit tests do
 * (test/unit/test_empty_zip.c, test/unit/test_set_file_compression.c, and the extract loop of
 * src/main.c:429-577), against libotezip_amd.so.
 *   zip_roundtrip list ARCHIVE            -> "n_entries\nname method comp uncomp crc\n..."
 *   zip_roundtrip extract ARCHIVE OUTDIR  -> writes every entry to OUTDIR/<index>, prints "i size|NULL"
 *   zip_roundtrip setcomp PATH            -> the test_set_file_compression scenario; exit 0 on success
 */
#include <otezip/zip.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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
	return 64;
}
