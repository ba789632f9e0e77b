/* zip_api.c -- the libzip-subset host API of otezip (include/otezip/zip.h) in
 * plain C on top of the ozg_* GPU ABI (include/otezip_gpu.h).
 *
 * Mirrors the reference's public functions one for one (ref
 * src/lib/otezip.c:693-785, :1079-1237, :1273-1440, :1592-1667) and its container
 * plumbing (EOCD / central directory reader :199-396, LFH/CDH/EOCD writer
 * :1443-1590).  What is NOT here is a codec: every payload goes through the GPU.
 *
 * The structural change is batching, on both sides.
 *
 * Reading (ref otezip_extract_entry :399-684, one entry per zip_fopen_index call): after
 * zip_open the whole directory is known, so entries are decoded a WINDOW at a time: the
 * window's stretch of the file is read with one read() into pinned memory (the local
 * headers are parsed from that buffer: no seek + 30-byte read per entry), the batch is
 * submitted asynchronously (H2D, kernels, D2H on a stream of its own) and the outputs land
 * in a pinned arena that zip_file_t.data points into.  Sequential callers get read-ahead:
 * while they consume window k, window k+1 is already being read, copied and decoded; with
 * several GPUs (OTEZIP_GPU_DEVICES) the windows go to the devices in turn.  The first window
 * after a miss is small and aligned to a grid (opening one file by name decodes 64 entries,
 * not thousands); sequential use grows the windows.  zip_file_t keeps the reference's public
 * prefix {data,size,pos}; zip_fclose releases a window when its last reader goes away.
 *
 * Writing (ref zip_file_add :1079-1183, one codec call per file): zip_file_add copies the
 * source into a pinned batch buffer and returns; the batch is compressed by ONE
 * ozg_encode_batch_host call when it is full, when the entry is needed (stat, open, replace)
 * or at zip_close, and the local headers + payloads are then written in index order.  Until
 * then zip_set_file_compression changes the method the entry is actually encoded with (the
 * libzip meaning of the call; the reference relabels bytes that are already on disk).
 */
#define _GNU_SOURCE
#include <otezip/zip.h>
#include <otezip_gpu.h>

#include <pthread.h>
#include <sys/stat.h>
#include <unistd.h>

#define SIG_LFH 0x04034b50u
#define SIG_CDH 0x02014b50u
#define SIG_EOCD 0x06054b50u
#define MAX_PAYLOAD (2ULL * 1024ULL * 1024ULL * 1024ULL) /* ref otezip.c:102 */
#define MAX_FIELD_LEN (64u * 1024u - 1u)

/* window limits: entries, compressed bytes, output bytes per GPU batch */
#define WIN_FIRST_ENTRIES 64u /* first window after a miss (a power of two: windows start on this grid) */
#define WIN_MAX_ENTRIES 1024u /* (pinning host memory costs ~0.25 ms per MB: windows stay small enough to recycle three sets) */
#define WIN_MAX_COMP (48ull << 20)
#define WIN_MAX_OUT (96ull << 20)
#define READ_THREADS_MAX 16
static int g_read_threads = 4; /* a window's stretch of the file is read by this many pread()s in parallel (OTEZIP_READ_THREADS) */
static int g_ahead_min = 2;    /* windows kept in preparation behind the current one, at least (OTEZIP_READ_AHEAD).  A window's way to
                                * its D2H copy -- read 0.6 ms, H2D 0.5, kernels 0.35 -- is longer than the copy itself (1.3 ms for
                                * 67 MB): with one window ahead the copy engine idles 0.3 ms per window, with two (submitted in
                                * order, copies first come first served) it does not (tools/api_rates.sh, tools/api_trace.sh) */
#define AHEAD_MAX 8    /* windows prepared ahead of a sequential reader: one per device (OTEZIP_GPU_DEVICES), at most this many */
#define FLIGHTS_MAX (AHEAD_MAX + 1) /* encode batches in flight while a flush writes: one per device + the one being written */
/* Pending adds are flushed at this many source bytes / entries: the ZIP32 limits, so that in
 * practice a batch lives until zip_close (or until one of its entries is needed) and
 * zip_set_file_compression reaches every entry added since. */
#define PEND_FLUSH_BYTES (3ull << 30)
#define PEND_FLUSH_ITEMS 65535u
#define PCHUNK_BYTES ((size_t)32 << 20) /* sources per encode batch (an entry larger than this gets a chunk of its own) */
#define PCHUNK_ITEMS 16384u

int otezip_verify_crc = 0;
uint64_t otezip_max_expansion_ratio = 1000ULL;
uint64_t otezip_max_expansion_slack = 1024ULL * 1024ULL;
int otezip_ignore_zipbomb = 0;

/* pinned host memory of a window; recycled between windows (pinning is expensive) */
struct winbuf {
	uint8_t *src, *out;
	ozg_result *res;
	size_t src_cap, out_cap, res_cap;
};

struct window {
	zip_uint64_t first, count; /* entries [first, first+count) */
	struct winbuf b;           /* source bytes, outputs (entry k at b.out + off[k]), batch results */
	uint64_t *off;
	ozg_result *res;           /* per entry */
	uint8_t *valid;            /* 0: failed the pre-checks of ref :411-446 */
	uint32_t *slot;            /* position of entry k in the submitted batch */
	ozg_job *job;              /* in flight; NULL once waited for (or nothing was submitted) */
	int failed;
	long readers;              /* open zip_file_t that point into b.out */
	int detached;              /* no longer one of the archive's windows */
	struct ozh_zip *owner;
	/* a read-ahead window is filled (file read, local headers, submit) by a helper thread */
	pthread_t th;
	int th_live;
	int quiet;                 /* read-ahead: a failure is not reported, the reader starts over when it gets there */
	struct win_fill *fill;
};

struct pend_item {
	zip_uint64_t index;
	uint64_t off; /* inside its chunk */
	uint32_t len;
};

/* sources of added entries wait in plain heap chunks (pinning memory costs more than copying it
 * once more into a pinned buffer that is reused); a chunk is one encode batch */
struct pchunk {
	uint8_t *buf;
	size_t cap, used;
	size_t item0, nitems; /* its entries: pit[item0 .. item0 + nitems) */
};

struct ozh_zip {
	struct zip pub; /* MUST stay first: callers see this prefix */
	struct window *win;         /* current window */
	struct window *ahead[AHEAD_MAX]; /* the ones being prepared behind it, in index order */
	int n_ahead;
	zip_uint64_t last_index;
	unsigned seq_hits;     /* consecutive opens of index, index + 1, ... */
	zip_uint64_t grow;     /* size of the next read-ahead window */
	uint32_t cd_ofs;       /* start of the central directory when the archive was opened */
	int ndev, next_dev;
	unsigned long sub_next, sub_done; /* windows are submitted in the order they were begun (tickets: g_sub_mu) */
	/* deferred zip_file_add: sources, in index order */
	struct pchunk *pch;
	size_t npch, pchcap;
	size_t ptotal; /* bytes pending in all chunks */
	struct pend_item *pit;
	size_t pn, pncap;
	int write_failed;
};

struct ozh_file {
	struct zip_file pub; /* MUST stay first */
	struct window *win;  /* NULL: pub.data is a private heap copy */
};

#include <time.h>
static int g_trace = -1;
static double trace_now(void) {
	struct timespec ts;
	clock_gettime (CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
/* OTEZIP_TRACE=1: one line per pipeline step on stderr (diagnostics) */
#define TRACE(...) \
	do { \
		if (g_trace < 0) { \
			g_trace = getenv ("OTEZIP_TRACE") ? 1 : 0; \
		} \
		if (g_trace) { \
			fprintf (stderr, "[otezip-amd %.6f] ", trace_now ()); \
			fprintf (stderr, __VA_ARGS__); \
			fprintf (stderr, "\n"); \
		} \
	} while (0)

static int g_gpu_state; /* 0 untried, 1 ok, -1 failed */
static int g_ndev = 1, g_dev0 = 0;

/* OTEZIP_GPU_DEVICE = first device (default 0), OTEZIP_GPU_DEVICES = how many consecutive
 * devices the windows of an archive are dealt to (default 1); device numbers are the logical ones
 * of otezip_gpu.h (OTEZIP_GPU_DEVICE_MAP) */
static pthread_mutex_t g_gpu_mu = PTHREAD_MUTEX_INITIALIZER;
static int gpu_up(void) {
	pthread_mutex_lock (&g_gpu_mu);
	if (g_gpu_state == 0) {
		TRACE ("initialising the GPU");
		const char *dev = getenv ("OTEZIP_GPU_DEVICE"), *cnt = getenv ("OTEZIP_GPU_DEVICES");
		g_dev0 = dev ? atoi (dev) : 0;
		int want = cnt ? atoi (cnt) : 1, have = ozg_device_count ();
		if (want < 1) {
			want = 1;
		}
		if (have > 0 && g_dev0 + want > have) {
			want = have > g_dev0 ? have - g_dev0 : 1;
		}
		if (want > AHEAD_MAX) {
			want = AHEAD_MAX;
		}
		g_ndev = 0;
		/* first device first: it becomes the default of every thread that has not chosen one */
		for (int k = 0; k < want; k++) {
			int rc = ozg_init (g_dev0 + k);
			if (rc != OZG_OK) {
				fprintf (stderr, "otezip-amd: %s\n", ozg_last_error ());
				if (k == 0) {
					g_gpu_state = -1;
					pthread_mutex_unlock (&g_gpu_mu);
					return 0;
				}
				want = k;
				break;
			}
		}
		(void)ozg_use_device (g_dev0);
		{
			const char *rt = getenv ("OTEZIP_READ_THREADS"), *ra = getenv ("OTEZIP_READ_AHEAD");
			if (rt && atoi (rt) >= 1) {
				g_read_threads = atoi (rt) > READ_THREADS_MAX ? READ_THREADS_MAX : atoi (rt);
			}
			if (ra && atoi (ra) >= 1) {
				g_ahead_min = atoi (ra) > AHEAD_MAX ? AHEAD_MAX : atoi (ra);
			}
		}
		g_ndev = want;
		g_gpu_state = 1;
		TRACE ("devices ready: %d from %d", g_ndev, g_dev0);
	}
	const int up = g_gpu_state == 1;
	pthread_mutex_unlock (&g_gpu_mu);
	return up;
}

/* Creating the HIP context takes a quarter of a second: a writer starts it in the background at
 * its first zip_file_add, so that it overlaps with the adds (whoever needs the GPU first waits
 * on the mutex inside gpu_up). */
static void *gpu_up_thread(void *arg) {
	(void)arg;
	(void)gpu_up ();
	return NULL;
}
static void gpu_up_in_background(void) {
	static int started;
	if (g_gpu_state != 0 || __atomic_exchange_n (&started, 1, __ATOMIC_ACQ_REL)) {
		return;
	}
	pthread_t t;
	if (pthread_create (&t, NULL, gpu_up_thread, NULL) == 0) {
		pthread_detach (t);
	}
}

/* ------------------------------------------------------------ small helpers */

static uint16_t rd16(const uint8_t *p) { return otezip_read_le16 (p); }
static uint32_t rd32(const uint8_t *p) { return otezip_read_le32 (p); }

static int read_fully(FILE *fp, void *dst, size_t n) { return fread (dst, 1, n, fp) == n ? 0 : -1; }

static void dostime_now(uint16_t *t, uint16_t *d) {
	/* same encoding and clamping as ref src/lib/time.inc.c:29-70 */
	time_t now = time (NULL);
	struct tm tmv;
	if (now == (time_t)-1 || !localtime_r (&now, &tmv)) {
		*t = 0;
		*d = (uint16_t) ((1 << 5) | 1);
		return;
	}
	int yo = tmv.tm_year + 1900 - 1980;
	yo = yo < 0 ? 0 : yo > 127 ? 127 : yo;
	*t = (uint16_t) ((tmv.tm_hour << 11) | (tmv.tm_min << 5) | ((tmv.tm_sec > 59 ? 59 : tmv.tm_sec) / 2));
	*d = (uint16_t) ((yo << 9) | ((tmv.tm_mon + 1) << 5) | tmv.tm_mday);
}

int otezip_method_from_string(const char *name) {
	static const struct {
		const char *n;
		int m;
	} tab[] = { { "store", OTEZIP_METHOD_STORE }, { "deflate", OTEZIP_METHOD_DEFLATE }, { "zstd", OTEZIP_METHOD_ZSTD },
		{ "lzma", OTEZIP_METHOD_LZMA }, { "brotli", OTEZIP_METHOD_BROTLI }, { "lzfse", OTEZIP_METHOD_LZFSE } };
	if (!name) {
		return -1;
	}
	for (size_t i = 0; i < sizeof (tab) / sizeof (tab[0]); i++) {
		if (!strcmp (name, tab[i].n)) {
			return tab[i].m;
		}
	}
	return -1;
}

/* ------------------------------------------------ central directory reader */

#define ERR_READ (-1)
#define ERR_INCONS (-2)

/* ref otezip_find_eocd, otezip.c:199-272: last EOCD signature whose directory
 * lies inside the file and starts with a CDH signature */
static long find_eocd(FILE *fp, size_t *cd_size, uint32_t *cd_ofs, uint16_t *n_entries) {
	if (fseek (fp, 0, SEEK_END) != 0) {
		return ERR_READ;
	}
	long fsz = ftell (fp);
	if (fsz < 22) {
		return ERR_INCONS;
	}
	size_t span = fsz < 0x10000 + 22 ? (size_t)fsz : 0x10000 + 22;
	uint8_t *buf = (uint8_t *)malloc (span);
	if (!buf) {
		return ERR_READ;
	}
	if (fseek (fp, fsz - (long)span, SEEK_SET) != 0 || read_fully (fp, buf, span) != 0) {
		free (buf);
		return ERR_READ;
	}
	long found = ERR_INCONS;
	for (size_t i = span - 22 + 1; i-- > 0;) {
		if (rd32 (buf + i) != SIG_EOCD) {
			continue;
		}
		uint16_t n = rd16 (buf + i + 10);
		uint32_t sz = rd32 (buf + i + 12), ofs = rd32 (buf + i + 16);
		if (ofs > (uint32_t)fsz || (uint64_t)ofs + sz > (uint64_t)fsz) {
			continue;
		}
		if (n > 0 && sz >= 4) {
			uint8_t sig[4];
			if (fseek (fp, (long)ofs, SEEK_SET) != 0 || read_fully (fp, sig, 4) != 0 || rd32 (sig) != SIG_CDH) {
				continue;
			}
		}
		*n_entries = n;
		*cd_size = sz;
		*cd_ofs = ofs;
		found = fsz - (long)span + (long)i;
		break;
	}
	free (buf);
	return found;
}

/* ref otezip_load_central, otezip.c:275-396 */
static int load_central(zip_t *za) {
	size_t cd_size = 0;
	uint32_t cd_ofs = 0;
	uint16_t n = 0;
	long pos = find_eocd (za->fp, &cd_size, &cd_ofs, &n);
	if (pos < 0) {
		return (int)pos;
	}
	((struct ozh_zip *)za)->cd_ofs = cd_ofs;
	if (n == 0) {
		if (cd_size != 0) {
			return ERR_INCONS;
		}
		za->entries = NULL;
		za->n_entries = 0;
		return 0;
	}
	if (cd_size > MAX_PAYLOAD || (size_t)n * 46 > cd_size) {
		return ERR_INCONS;
	}
	uint8_t *cd = (uint8_t *)malloc (cd_size);
	if (!cd) {
		return ERR_READ;
	}
	if (fseek (za->fp, (long)cd_ofs, SEEK_SET) != 0 || read_fully (za->fp, cd, cd_size) != 0) {
		free (cd);
		return ERR_READ;
	}
	za->entries = (struct otezip_entry *)calloc (n, sizeof (struct otezip_entry));
	if (!za->entries) {
		free (cd);
		return ERR_READ;
	}
	za->n_entries = n;
	size_t off = 0;
	for (uint16_t i = 0; i < n; i++) {
		if (off + 46 > cd_size || rd32 (cd + off) != SIG_CDH) {
			free (cd);
			return ERR_INCONS;
		}
		const uint8_t *h = cd + off;
		size_t nlen = rd16 (h + 28), total = 46 + nlen + rd16 (h + 30) + rd16 (h + 32);
		if (total > cd_size - off) {
			free (cd);
			return ERR_INCONS;
		}
		struct otezip_entry *e = &za->entries[i];
		e->method = rd16 (h + 10);
		e->file_time = rd16 (h + 12);
		e->file_date = rd16 (h + 14);
		e->crc32 = rd32 (h + 16);
		e->comp_size = rd32 (h + 20);
		e->uncomp_size = rd32 (h + 24);
		e->external_attr = rd32 (h + 38);
		e->local_hdr_ofs = rd32 (h + 42);
		if (e->comp_size > MAX_PAYLOAD || e->uncomp_size > MAX_PAYLOAD) {
			free (cd);
			return ERR_INCONS;
		}
		e->name = (char *)malloc (nlen + 1);
		if (!e->name) {
			free (cd);
			return ERR_READ;
		}
		memcpy (e->name, h + 46, nlen);
		e->name[nlen] = 0;
		off += total;
	}
	free (cd);
	return 0;
}

/* ------------------------------------------------------------- open / close */

zip_t *zip_open(const char *path, int flags, int *errorp) {
	/* ref otezip.c:693-785 */
	struct ozh_zip *z = (struct ozh_zip *)calloc (1, sizeof (*z));
	if (!z) {
		if (errorp) {
			*errorp = -1;
		}
		return NULL;
	}
	zip_t *za = &z->pub;
	const char *mode = "rb";
	int exists = 0;
	if (flags & ZIP_CREATE) {
		if ((flags & ZIP_EXCL) && (flags & ZIP_TRUNCATE)) {
			if (errorp) {
				*errorp = -1;
			}
			free (z);
			return NULL;
		}
		FILE *t = fopen (path, "rb");
		exists = t != NULL;
		if (t) {
			fclose (t);
		}
		if (exists && (flags & ZIP_EXCL)) {
			if (errorp) {
				*errorp = -1;
			}
			free (z);
			return NULL;
		}
		mode = (exists && !(flags & ZIP_TRUNCATE)) ? "r+b" : "w+b";
		za->mode = 1;
	}
	if (mode[0] == 'w') {
		unlink (path);
	}
	za->fp = fopen (path, mode);
	if (!za->fp) {
		if (errorp) {
			*errorp = ZIP_ER_OPEN;
		}
		free (z);
		return NULL;
	}
	if (za->mode == 0 || (exists && !(flags & ZIP_TRUNCATE))) {
		int rc = load_central (za);
		if (rc != 0) {
			if (errorp) {
				*errorp = rc == ERR_READ ? ZIP_ER_READ : rc == ERR_INCONS ? ZIP_ER_INCONS : ZIP_ER_NOZIP;
			}
			zip_close (za);
			return NULL;
		}
		if (za->mode == 1) {
			za->next_index = za->n_entries;
			/* append: new local headers go where the old central directory started.
			 * (the reference keeps writing after it, ref :758-779, leaving a dead
			 * directory in the middle of the file; both layouts are valid ZIP) */
			fseek (za->fp, 0, SEEK_END);
		}
	}
	if (errorp) {
		*errorp = 0;
	}
	return za;
}

zip_t *zip_open_from_source(zip_source_t *src, int flags, zip_error_t *error) {
	/* ref otezip.c:1406-1440: spill the buffer to a temp file, open that */
	(void)error;
	if (!src) {
		return NULL;
	}
	char tmp[] = "/tmp/otezip_XXXXXX";
	mode_t um = umask (077);
	int fd = mkstemp (tmp);
	umask (um);
	if (fd < 0) {
		return NULL;
	}
	ssize_t w = write (fd, src->buf, (size_t)src->len);
	close (fd);
	if (w < 0 || (zip_uint64_t)w != src->len) {
		unlink (tmp);
		return NULL;
	}
	int err = 0;
	zip_t *za = zip_open (tmp, flags, &err);
	if (!za) {
		unlink (tmp);
	}
	return za;
}

/* ------------------------------------------------------------------ windows */

static void winbuf_free(struct winbuf *b) {
	ozg_free_pinned (b->src);
	ozg_free_pinned (b->out);
	ozg_free_pinned (b->res);
	memset (b, 0, sizeof (*b));
}

/* Pinned buffer sets outlive their windows and their archives: a process-wide pool (pinning
 * is the expensive part of the whole ingest path).  Guarded by a mutex: archives may be used
 * from different threads. */
#define POOL_SETS (AHEAD_MAX + 4)
static struct winbuf g_pool[POOL_SETS];
static pthread_mutex_t g_pool_mu = PTHREAD_MUTEX_INITIALIZER;

/* give a pinned set back: kept for the next user if there is room, or if it is larger than a kept one */
static void winbuf_put(struct winbuf *b) {
	if (!b->src && !b->out && !b->res) {
		return;
	}
	int kept = 0;
	pthread_mutex_lock (&g_pool_mu);
	for (int i = 0; i < POOL_SETS; i++) {
		if (!g_pool[i].src && !g_pool[i].out && !g_pool[i].res) {
			g_pool[i] = *b;
			kept = 1;
			break;
		}
	}
	if (!kept) { /* full: keep the larger sets */
		int small = 0;
		for (int i = 1; i < POOL_SETS; i++) {
			if (g_pool[i].out_cap < g_pool[small].out_cap) {
				small = i;
			}
		}
		if (g_pool[small].out_cap < b->out_cap) {
			struct winbuf t = g_pool[small];
			g_pool[small] = *b;
			*b = t;
		}
	}
	pthread_mutex_unlock (&g_pool_mu);
	if (!kept) {
		winbuf_free (b);
	} else {
		memset (b, 0, sizeof (*b));
	}
}

static void window_join(struct window *w);
static void window_free(struct window *w) {
	if (!w) {
		return;
	}
	window_join (w);
	if (w->job) {
		ozg_job_wait (w->job); /* the copies still write into its buffers */
		w->job = NULL;
	}
	winbuf_put (&w->b);
	free (w->off);
	free (w->res);
	free (w->valid);
	free (w->slot);
	free (w);
}

static void window_detach(struct window *w) {
	if (!w) {
		return;
	}
	if (w->readers == 0) {
		window_free (w);
	} else {
		w->detached = 1;
	}
}

/* pinned memory of at least the given sizes: the smallest pooled set that is large enough */
static int winbuf_get(struct ozh_zip *z, struct winbuf *b, size_t src, size_t out, size_t res) {
	(void)z;
	memset (b, 0, sizeof (*b));
	int best = -1;
	pthread_mutex_lock (&g_pool_mu);
	for (int i = 0; i < POOL_SETS; i++) {
		if (g_pool[i].src_cap >= src && g_pool[i].out_cap >= out && g_pool[i].res_cap >= res &&
			(best < 0 || g_pool[i].out_cap < g_pool[best].out_cap)) {
			best = i;
		}
	}
	if (best >= 0) {
		*b = g_pool[best];
		memset (&g_pool[best], 0, sizeof (g_pool[best]));
	}
	pthread_mutex_unlock (&g_pool_mu);
	if (best >= 0) {
		return 0;
	}
	/* none fits: sizes are rounded up so that the sets of one archive's windows are interchangeable */
	b->src_cap = (src + (src >> 2) + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1u);
	b->out_cap = (out + (out >> 2) + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1u);
	b->res_cap = res < WIN_MAX_ENTRIES ? WIN_MAX_ENTRIES : res;
	b->src = (uint8_t *)ozg_malloc_pinned (b->src_cap);
	b->out = (uint8_t *)ozg_malloc_pinned (b->out_cap);
	b->res = (ozg_result *)ozg_malloc_pinned (b->res_cap * sizeof (ozg_result));
	if (!b->src || !b->out || !b->res) {
		winbuf_free (b);
		return -1;
	}
	return 0;
}

/* ---- parallel read of one stretch of the file (page-cache copies run at a few GB/s per thread) */
struct pread_job {
	int fd;
	uint8_t *dst;
	uint64_t off, len;
	int ok;
};
static void *pread_run(void *p) {
	struct pread_job *j = (struct pread_job *)p;
	uint64_t done = 0;
	j->ok = 1;
	while (done < j->len) {
		ssize_t r = pread (j->fd, j->dst + done, (size_t) (j->len - done), (off_t) (j->off + done));
		if (r <= 0) {
			j->ok = 0;
			break;
		}
		done += (uint64_t)r;
	}
	return NULL;
}
static int read_span(FILE *fp, uint8_t *dst, uint64_t off, uint64_t len) {
	const int fd = fileno (fp);
	int nt = len >= ((uint64_t)4 << 20) ? g_read_threads : 1;
	struct pread_job job[READ_THREADS_MAX];
	pthread_t th[READ_THREADS_MAX];
	int started[READ_THREADS_MAX];
	const uint64_t piece = ((len + (uint64_t)nt - 1u) / (uint64_t)nt + 4095u) & ~(uint64_t)4095u;
	for (int i = 0; i < nt; i++) {
		const uint64_t o = piece * (uint64_t)i;
		job[i].fd = fd;
		job[i].dst = dst + o;
		job[i].off = off + o;
		job[i].len = o >= len ? 0 : (len - o < piece ? len - o : piece);
		job[i].ok = 1;
		started[i] = 0;
	}
	for (int i = 1; i < nt; i++) {
		started[i] = job[i].len && pthread_create (&th[i], NULL, pread_run, &job[i]) == 0;
	}
	pread_run (&job[0]);
	int ok = job[0].ok;
	for (int i = 1; i < nt; i++) {
		if (started[i]) {
			pthread_join (th[i], NULL);
		} else if (job[i].len) {
			pread_run (&job[i]); /* no thread: read it here */
		}
		ok = ok && job[i].ok;
	}
	return ok ? 0 : -1;
}

/* memcpy of a large block by the reader's threads (a chunk of sources on its way into pinned memory) */
struct copy_job {
	uint8_t *dst;
	const uint8_t *src;
	size_t len;
};
static void *copy_run(void *p) {
	const struct copy_job *j = (const struct copy_job *)p;
	memcpy (j->dst, j->src, j->len);
	return NULL;
}
static void copy_span(uint8_t *dst, const uint8_t *src, size_t len) {
	const int nt = len >= ((size_t)4 << 20) ? g_read_threads : 1;
	struct copy_job job[READ_THREADS_MAX];
	pthread_t th[READ_THREADS_MAX];
	int started[READ_THREADS_MAX];
	const size_t piece = ((len + (size_t)nt - 1u) / (size_t)nt + 4095u) & ~(size_t)4095u;
	for (int i = 0; i < nt; i++) {
		const size_t o = piece * (size_t)i;
		job[i].dst = dst + o;
		job[i].src = src + o;
		job[i].len = o >= len ? 0 : (len - o < piece ? len - o : piece);
		started[i] = i > 0 && job[i].len && pthread_create (&th[i], NULL, copy_run, &job[i]) == 0;
	}
	copy_run (&job[0]);
	for (int i = 1; i < nt; i++) {
		if (started[i]) {
			pthread_join (th[i], NULL);
		} else if (job[i].len) {
			copy_run (&job[i]);
		}
	}
}

static void put16(uint8_t *p, uint16_t v) { otezip_write_le16 (p, v); }
static void put32(uint8_t *p, uint32_t v) { otezip_write_le32 (p, v); }

static uint32_t write_lfh(FILE *fp, const struct otezip_entry *e) {
	/* ref otezip.c:1443-1491 */
	size_t nl = strlen (e->name);
	if (nl > MAX_FIELD_LEN) {
		nl = MAX_FIELD_LEN;
	}
	uint8_t h[30];
	put32 (h, SIG_LFH);
	put16 (h + 4, 20);
	put16 (h + 6, 0);
	put16 (h + 8, e->method);
	put16 (h + 10, e->file_time);
	put16 (h + 12, e->file_date);
	put32 (h + 14, e->crc32);
	put32 (h + 18, e->comp_size);
	put32 (h + 22, e->uncomp_size);
	put16 (h + 26, (uint16_t)nl);
	put16 (h + 28, 0);
	fwrite (h, 1, 30, fp);
	fwrite (e->name, 1, nl, fp);
	return 30 + (uint32_t)nl;
}

static int finalize(zip_t *za) {
	/* ref otezip.c:1240-1271, :1494-1590 */
	if (fseek (za->fp, 0, SEEK_END) != 0) { /* (entries may have been read back since the last write) */
		return -1;
	}
	long cd_ofs = ftell (za->fp);
	if (cd_ofs < 0 || (uint64_t)cd_ofs > UINT32_MAX) {
		return -1;
	}
	uint64_t cd_size = 0;
	for (zip_uint64_t i = 0; i < za->n_entries; i++) {
		const struct otezip_entry *e = &za->entries[i];
		size_t nl = strlen (e->name);
		if (nl > MAX_FIELD_LEN) {
			nl = MAX_FIELD_LEN;
		}
		uint8_t h[46];
		memset (h, 0, sizeof (h));
		put32 (h, SIG_CDH);
		put16 (h + 4, 0x031e);
		put16 (h + 6, 20);
		put16 (h + 10, e->method);
		put16 (h + 12, e->file_time);
		put16 (h + 14, e->file_date);
		put32 (h + 16, e->crc32);
		put32 (h + 20, e->comp_size);
		put32 (h + 24, e->uncomp_size);
		put16 (h + 28, (uint16_t)nl);
		put32 (h + 38, e->external_attr);
		put32 (h + 42, e->local_hdr_ofs);
		fwrite (h, 1, 46, za->fp);
		fwrite (e->name, 1, nl, za->fp);
		cd_size += 46 + nl;
		if (cd_size > UINT32_MAX) {
			return -1;
		}
	}
	uint8_t eocd[22];
	memset (eocd, 0, sizeof (eocd));
	put32 (eocd, SIG_EOCD);
	put16 (eocd + 8, (uint16_t)za->n_entries);
	put16 (eocd + 10, (uint16_t)za->n_entries);
	put32 (eocd + 12, (uint32_t)cd_size);
	put32 (eocd + 16, (uint32_t)cd_ofs);
	fwrite (eocd, 1, 22, za->fp);
	return 0;
}

static int flush_pending(zip_t *za);
static int flush_pending_ex(zip_t *za, int is_replace);
static void pending_drop(struct ozh_zip *z);

int zip_close(zip_t *za) {
	/* ref otezip.c:1273-1295 */
	if (!za || !za->fp) {
		free (za);
		return -1;
	}
	struct ozh_zip *z = (struct ozh_zip *)za;
	int rc = 0;
	for (int k = 0; k < z->n_ahead; k++) {
		window_join (z->ahead[k]); /* (a helper thread reads this archive's directory and file) */
	}
	if (za->mode == 1) {
		if (flush_pending (za) != 0 || z->write_failed) {
			rc = -1;
		}
		if (finalize (za) != 0) {
			rc = -1;
		}
	}
	if (fclose (za->fp) != 0 && za->mode == 1) {
		rc = -1;
	}
	for (zip_uint64_t i = 0; i < za->n_entries; i++) {
		free (za->entries[i].name);
	}
	free (za->entries);
	if (z->win) {
		z->win->owner = NULL;
	}
	window_detach (z->win);
	for (int k = 0; k < z->n_ahead; k++) {
		z->ahead[k]->owner = NULL;
		window_detach (z->ahead[k]);
	}
	pending_drop (z);
	free (z->pch);
	free (z->pit);
	free (z);
	return rc;
}

/* --------------------------------------------------------------- directory */

static int valid(const zip_t *za) { return za != NULL && za->fp != NULL; }

/* added, not yet encoded and written: adds only append, so the pending entries are the tail */
static int is_pending(const zip_t *za, zip_uint64_t index) {
	const struct ozh_zip *z = (const struct ozh_zip *)za;
	return z->pn > 0 && index >= z->pit[0].index;
}

zip_uint64_t zip_get_num_files(zip_t *za) { return za ? za->n_entries : 0u; }

zip_int64_t zip_name_locate(zip_t *za, const char *fname, zip_flags_t flags) {
	(void)flags;
	if (!valid (za) || !fname) {
		return -1;
	}
	for (zip_uint64_t i = 0; i < za->n_entries; i++) {
		if (!strcmp (za->entries[i].name, fname)) {
			return (zip_int64_t)i;
		}
	}
	return -1;
}

const char *zip_get_name(zip_t *za, zip_uint64_t index, zip_flags_t flags) {
	(void)flags;
	return (valid (za) && index < za->n_entries) ? za->entries[index].name : NULL;
}

void zip_stat_init(zip_stat_t *st) {
	if (st) {
		memset (st, 0, sizeof (*st));
		st->index = ZIP_UINT64_MAX;
		st->mtime = (time_t)-1;
	}
}

int zip_stat_index(zip_t *za, zip_uint64_t index, zip_flags_t flags, zip_stat_t *st) {
	(void)flags;
	if (!valid (za) || !st || index >= za->n_entries) {
		return -1;
	}
	if (za->mode == 1 && is_pending (za, index) && flush_pending (za) != 0) {
		return -1; /* (sizes and CRC of an added entry exist once its batch has been encoded) */
	}
	zip_stat_init (st);
	const struct otezip_entry *e = &za->entries[index];
	st->name = e->name;
	st->index = index;
	st->size = e->uncomp_size;
	st->comp_size = e->comp_size;
	st->crc = e->crc32;
	st->comp_method = e->method;
	st->valid = ZIP_STAT_NAME | ZIP_STAT_INDEX | ZIP_STAT_SIZE | ZIP_STAT_COMP_SIZE | ZIP_STAT_CRC | ZIP_STAT_COMP_METHOD;
	return 0;
}

int zip_stat(zip_t *za, const char *fname, zip_flags_t flags, zip_stat_t *st) {
	zip_int64_t i = zip_name_locate (za, fname, flags);
	return i < 0 ? -1 : zip_stat_index (za, (zip_uint64_t)i, flags, st);
}

/* ------------------------------------------------------------------ extract */

/* ref otezip.c:454-462, on the host: an entry the guard rejects gets no output space and is
 * never sent to the device (a 100-byte archive must not cost gigabytes of memory) */
static int host_zipbomb(const struct otezip_entry *e) {
	if (otezip_ignore_zipbomb || e->comp_size == 0) {
		return 0;
	}
	return (uint64_t)e->uncomp_size > (uint64_t)e->comp_size * otezip_max_expansion_ratio + otezip_max_expansion_slack;
}

/* what filling a window needs once it is planned: the file stretch, the batch under construction */
struct win_fill {
	zip_t *za;
	uint64_t fsz, lo, hi, out;
	int one_span;
	int dev;
	ozg_entry *ents;
	unsigned long ticket;
	int ticket_held;
};

/* The copies of a device's jobs run first come, first served (otz_kernels.hip), so the order of submission is
 * the order of completion: windows filled by helper threads must not overtake each other on the way to the
 * GPU, or the one the reader waits for queues behind the one after it.  A window takes a ticket when it is
 * begun, waits for its turn right before the submit and gives the turn on whatever happens. */
static pthread_mutex_t g_sub_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_sub_cv = PTHREAD_COND_INITIALIZER;

static void ticket_take(struct ozh_zip *z, struct win_fill *F) {
	pthread_mutex_lock (&g_sub_mu);
	F->ticket = z->sub_next++;
	F->ticket_held = 1;
	pthread_mutex_unlock (&g_sub_mu);
}
static void ticket_wait(struct ozh_zip *z, const struct win_fill *F) {
	pthread_mutex_lock (&g_sub_mu);
	while (F->ticket_held && z->sub_done != F->ticket) {
		pthread_cond_wait (&g_sub_cv, &g_sub_mu);
	}
	pthread_mutex_unlock (&g_sub_mu);
}
static void ticket_release(struct ozh_zip *z, struct win_fill *F) {
	if (!F->ticket_held) {
		return;
	}
	pthread_mutex_lock (&g_sub_mu);
	while (z->sub_done != F->ticket) { /* (left before its turn -- an early failure: the turn still passes in order) */
		pthread_cond_wait (&g_sub_cv, &g_sub_mu);
	}
	z->sub_done = F->ticket + 1u;
	F->ticket_held = 0;
	pthread_cond_broadcast (&g_sub_cv);
	pthread_mutex_unlock (&g_sub_mu);
}

/* Second half of window_begin: read the window's bytes, check the local headers (ref otezip.c:411-446),
 * submit the batch.  Runs on the caller's thread, or on a helper thread for a read-ahead window of a
 * read-only archive (then only pread() on the descriptor is used: the FILE's position belongs to the
 * caller).  Returns 0, -1, or 1: "the layout is not what the one-read path assumes, take the packed road". */
static int window_fill_body(struct window *w) {
	struct win_fill *F = w->fill;
	zip_t *za = F->za;
	const zip_uint64_t first = w->first, count = w->count;
	const uint64_t fsz = F->fsz, lo = F->lo, hi = F->hi;
	const int one_span = F->one_span;
	if (one_span && read_span (za->fp, w->b.src, lo, hi - lo) != 0) {
		return -1;
	}
	TRACE ("window [%llu): bytes read", (unsigned long long)first);
	uint32_t nb = 0;
	uint64_t o = 0, packed = 0;
	for (zip_uint64_t k = 0; k < count; k++) {
		const struct otezip_entry *e = &za->entries[first + k];
		w->off[k] = o;
		w->res[k].status = OZG_ST_INTERNAL;
		if (host_zipbomb (e)) {
			w->valid[k] = 1;
			w->slot[k] = UINT32_MAX;
			w->res[k].status = OZG_ST_BOMB;
			continue;
		}
		o += ((uint64_t)e->uncomp_size + 15u) & ~15ull;
		/* the checks of ref otezip.c:411-446 */
		if (e->comp_size > MAX_PAYLOAD || e->uncomp_size > MAX_PAYLOAD) {
			continue;
		}
		uint8_t lfh_buf[30];
		const uint8_t *lfh = lfh_buf;
		if (one_span) {
			if ((uint64_t)e->local_hdr_ofs + 30u > hi) {
				continue;
			}
			lfh = w->b.src + (e->local_hdr_ofs - lo);
		} else if ((uint64_t)e->local_hdr_ofs > fsz || fseek (za->fp, (long)e->local_hdr_ofs, SEEK_SET) != 0 ||
			read_fully (za->fp, lfh_buf, 30) != 0) {
			continue;
		}
		if (rd32 (lfh) != SIG_LFH) {
			continue;
		}
		const uint64_t d = (uint64_t)e->local_hdr_ofs + 30u + rd16 (lfh + 26) + rd16 (lfh + 28);
		if (d > fsz || d + e->comp_size > fsz) {
			continue;
		}
		uint64_t src_off;
		if (one_span) {
			if (d + e->comp_size > hi) {
				return 1; /* the payload reaches past the stretch that was read */
			}
			src_off = d - lo;
		} else {
			if (e->comp_size && (fseek (za->fp, (long)d, SEEK_SET) != 0 || read_fully (za->fp, w->b.src + packed, e->comp_size) != 0)) {
				continue;
			}
			src_off = packed;
			packed += ((uint64_t)e->comp_size + 15u) & ~15ull;
		}
		w->valid[k] = 1;
		w->slot[k] = nb;
		ozg_entry *b = &F->ents[nb++];
		b->src_off = src_off;
		b->dst_off = w->off[k];
		b->comp_size = e->comp_size;
		b->uncomp_size = e->uncomp_size;
		b->crc32 = e->crc32;
		b->method = e->method;
		b->flags = otezip_ignore_zipbomb ? OZG_EF_IGNORE_ZIPBOMB : 0;
	}
	if (nb > 0) {
		ozg_limits lim = { otezip_max_expansion_ratio, otezip_max_expansion_slack };
		ozg_set_limits (&lim);
		const uint64_t src_len = (one_span ? hi - lo : packed) + 16u, dst_len = F->out + 16u;
		ticket_wait ((struct ozh_zip *)za, F);
		int rc = ozg_use_device (F->dev);
		if (rc == OZG_OK) {
			rc = ozg_decode_submit (w->b.src, src_len, w->b.out, dst_len, F->ents, w->b.res, nb, &w->job);
		}
		ticket_release ((struct ozh_zip *)za, F);
		if (rc == OZG_E_NOMEM) {
			/* every staged-batch slot of the device is taken (other archives walk their windows too):
			 * back-pressure, not an error.  A read-ahead window gives up quietly -- the reader starts
			 * over when it gets there -- and a window somebody waits for takes the synchronous call. */
			if (w->quiet) {
				return -1;
			}
			TRACE ("window [%llu): no free slot, synchronous batch", (unsigned long long)first);
			w->job = NULL;
			rc = ozg_decode_batch_host (w->b.src, src_len, w->b.out, dst_len, F->ents, w->b.res, nb);
			if (rc == OZG_OK) {
				for (zip_uint64_t k = 0; k < count; k++) {
					if (w->valid[k] && w->slot[k] != UINT32_MAX) {
						w->res[k] = w->b.res[w->slot[k]];
					}
				}
			}
		}
		if (rc != OZG_OK) {
			if (!w->quiet) {
				fprintf (stderr, "otezip-amd: GPU decode failed: %s\n", ozg_last_error ());
			}
			return -1;
		}
	}
	TRACE ("window [%llu): submitted %u entries", (unsigned long long)first, nb);
	return 0;
}

static int window_fill(struct window *w) {
	const int rc = window_fill_body (w);
	ticket_release ((struct ozh_zip *)w->fill->za, w->fill);
	return rc;
}

static void win_fill_free(struct window *w) {
	if (w->fill) {
		free (w->fill->ents);
		free (w->fill);
		w->fill = NULL;
	}
}

static void *window_fill_thread(void *p) {
	struct window *w = (struct window *)p;
	if (window_fill (w) != 0) {
		w->failed = 1;
	}
	return NULL;
}

/* the helper thread of a read-ahead window is done after this */
static void window_join(struct window *w) {
	if (w->th_live) {
		pthread_join (w->th, NULL);
		w->th_live = 0;
	}
	win_fill_free (w);
}

/* Start decoding entries [first, first + up to want): read their bytes, submit the batch.
 * ahead: a read-ahead window -- filled by a helper thread where that is safe, quiet about failures.
 * Returns the window with its job in flight (window_wait completes it), or NULL. */
static struct window *window_begin_ex(zip_t *za, zip_uint64_t first, zip_uint64_t want, int packed_only, int ahead) {
	struct ozh_zip *z = (struct ozh_zip *)za;
	if (za->mode == 1) {
		fflush (za->fp);
	}
	uint64_t fsz;
	if (za->mode == 0) {
		struct stat sb; /* (a helper thread may be reading: leave the FILE's position alone) */
		if (fstat (fileno (za->fp), &sb) != 0) {
			return NULL;
		}
		fsz = (uint64_t)sb.st_size;
	} else {
		if (fseek (za->fp, 0, SEEK_END) != 0) {
			return NULL;
		}
		long fsl = ftell (za->fp);
		if (fsl < 0) {
			return NULL;
		}
		fsz = (uint64_t)fsl;
	}
	/* how far does the window reach */
	zip_uint64_t count = 0;
	uint64_t comp = 0, out = 0;
	while (first + count < za->n_entries && count < want && count < WIN_MAX_ENTRIES && !is_pending (za, first + count)) {
		const struct otezip_entry *e = &za->entries[first + count];
		uint64_t c = e->comp_size, o = host_zipbomb (e) ? 0 : ((uint64_t)e->uncomp_size + 15u) & ~15ull;
		if (count > 0 && (comp + c > WIN_MAX_COMP || out + o > WIN_MAX_OUT)) {
			break;
		}
		comp += c;
		out += o;
		count++;
	}
	if (count == 0) {
		return NULL;
	}
	struct window *w = (struct window *)calloc (1, sizeof (*w));
	struct win_fill *F = (struct win_fill *)calloc (1, sizeof (*F));
	if (!w || !F) {
		free (w);
		free (F);
		return NULL;
	}
	w->fill = F;
	w->owner = z;
	w->first = first;
	w->count = count;
	w->quiet = ahead;
	F->za = za;
	F->ents = (ozg_entry *)calloc (count, sizeof (ozg_entry));
	w->off = (uint64_t *)calloc (count, sizeof (uint64_t));
	w->res = (ozg_result *)calloc (count, sizeof (ozg_result));
	w->valid = (uint8_t *)calloc (count, 1);
	w->slot = (uint32_t *)calloc (count, sizeof (uint32_t));
	if (!F->ents || !w->off || !w->res || !w->valid || !w->slot) {
		goto fail;
	}
	/* Entries written in index order sit back to back: then ONE read covers local headers and
	 * payloads of the window, and the headers are parsed from the buffer.  Anything else (entries
	 * rewritten by zip_file_replace, foreign layouts) is read entry by entry into a packed buffer. */
	uint64_t lo = za->entries[first].local_hdr_ofs, hi;
	int one_span = !packed_only;
	for (zip_uint64_t k = 1; k < count && one_span; k++) {
		one_span = za->entries[first + k].local_hdr_ofs > za->entries[first + k - 1].local_hdr_ofs;
	}
	if (first + count < za->n_entries && !is_pending (za, first + count)) {
		hi = za->entries[first + count].local_hdr_ofs;
	} else {
		hi = za->mode == 0 && z->cd_ofs ? z->cd_ofs : fsz;
	}
	if (hi > fsz) {
		hi = fsz;
	}
	if (one_span && (hi <= za->entries[first + count - 1].local_hdr_ofs || hi - lo > comp + count * 2048ull + (1u << 20))) {
		one_span = 0;
	}
	const size_t src_need = (size_t) (one_span ? hi - lo : comp + 16u * count) + 64;
	TRACE ("window [%llu, +%llu): %s, %llu B in, %llu B out%s", (unsigned long long)first, (unsigned long long)count,
		one_span ? "one read" : "entry by entry", (unsigned long long)src_need, (unsigned long long)out, ahead ? " (ahead)" : "");
	if (winbuf_get (z, &w->b, src_need, (size_t)out + 64, count) != 0) {
		if (!ahead) {
			fprintf (stderr, "otezip-amd: %s\n", ozg_last_error ());
		}
		goto fail;
	}
	F->fsz = fsz;
	F->lo = lo;
	F->hi = hi;
	F->out = out;
	F->one_span = one_span;
	/* the devices take the windows in turn */
	F->dev = g_dev0 + (z->next_dev++ % (g_ndev > 0 ? g_ndev : 1));
	ticket_take (z, F);
	if (ahead && one_span && za->mode == 0 && pthread_create (&w->th, NULL, window_fill_thread, w) == 0) {
		w->th_live = 1; /* the caller's thread goes on consuming the current window meanwhile */
		return w;
	}
	{
		const int rc = window_fill (w);
		if (rc == 1) {
			window_free (w);
			return window_begin_ex (za, first, want, 1, ahead);
		}
		if (rc != 0) {
			goto fail;
		}
	}
	win_fill_free (w);
	return w;
fail:
	window_free (w);
	return NULL;
}

static struct window *window_begin(zip_t *za, zip_uint64_t first, zip_uint64_t want) { return window_begin_ex (za, first, want, 0, 0); }

/* the window's outputs and results are in host memory after this */
static int window_wait(struct window *w) {
	window_join (w);
	if (w->failed) {
		return -1;
	}
	if (w->job) {
		TRACE ("window [%llu): waiting", (unsigned long long)w->first);
		int rc = ozg_job_wait (w->job);
		TRACE ("window [%llu): ready", (unsigned long long)w->first);
		w->job = NULL;
		if (rc != OZG_OK) {
			if (!w->quiet) {
				fprintf (stderr, "otezip-amd: GPU decode failed: %s\n", ozg_last_error ());
			}
			w->failed = 1;
			return -1;
		}
		for (zip_uint64_t k = 0; k < w->count; k++) {
			if (w->valid[k] && w->slot[k] != UINT32_MAX) {
				w->res[k] = w->b.res[w->slot[k]];
			}
		}
	}
	return 0;
}

static int in_window(const struct window *w, zip_uint64_t index) { return w && index >= w->first && index < w->first + w->count; }

/* forget every decoded window (the archive's bytes changed) */
static void windows_drop(struct ozh_zip *z) {
	window_detach (z->win);
	z->win = NULL;
	for (int k = 0; k < z->n_ahead; k++) {
		window_detach (z->ahead[k]);
	}
	z->n_ahead = 0;
	z->seq_hits = 0;
}

zip_file_t *zip_fopen_index(zip_t *za, zip_uint64_t index, zip_flags_t flags) {
	/* ref otezip.c:1315-1334 over otezip_extract_entry :399-684 */
	(void)flags;
	if (!valid (za) || index >= za->n_entries) {
		return NULL;
	}
	struct ozh_zip *z = (struct ozh_zip *)za;
	if (!gpu_up ()) {
		return NULL; /* no device, no decoder: said so on stderr */
	}
	if (za->mode == 1 && is_pending (za, index) && flush_pending (za) != 0) {
		return NULL;
	}
	z->seq_hits = (z->win || z->n_ahead) && index == z->last_index + 1 ? z->seq_hits + 1 : 0;
	z->last_index = index;
	struct window *w = NULL;
	for (int attempt = 0; attempt < 2 && !w; attempt++) {
		if (!in_window (z->win, index)) {
			int hit = -1;
			for (int k = 0; k < z->n_ahead && hit < 0; k++) {
				hit = in_window (z->ahead[k], index) ? k : -1;
			}
			if (hit >= 0) {
				/* the reader has arrived at a prepared window (the ones it skipped go) */
				window_detach (z->win);
				for (int k = 0; k < hit; k++) {
					window_detach (z->ahead[k]);
				}
				z->win = z->ahead[hit];
				z->n_ahead -= hit + 1;
				memmove (z->ahead, z->ahead + hit + 1, (size_t)z->n_ahead * sizeof (z->ahead[0]));
			} else {
				windows_drop (z);
				z->grow = WIN_FIRST_ENTRIES;
				const zip_uint64_t start = index & ~(zip_uint64_t) (WIN_FIRST_ENTRIES - 1u);
				z->win = window_begin (za, start, WIN_FIRST_ENTRIES);
				if (z->win && !in_window (z->win, index)) { /* (a byte limit ended the window before the entry) */
					window_detach (z->win);
					z->win = NULL;
				}
				if (!z->win && start != index) {
					z->win = window_begin (za, index, WIN_FIRST_ENTRIES);
				}
				if (!z->win) {
					z->win = window_begin (za, index, 1); /* (an odd layout: one entry at a time) */
				}
				if (!z->win) {
					return NULL;
				}
			}
		}
		/* a caller that walks the archive gets the next windows prepared while it uses this one: one per
		 * device, so that every GPU of OTEZIP_GPU_DEVICES has a window in flight */
		if (z->seq_hits >= 2) {
			int depth = g_ndev > g_ahead_min ? g_ndev : g_ahead_min; /* (one per device at least) */
			depth = depth < AHEAD_MAX ? depth : AHEAD_MAX;
			while (z->n_ahead < depth) {
				const struct window *last = z->n_ahead ? z->ahead[z->n_ahead - 1] : z->win;
				const zip_uint64_t nf = last->first + last->count;
				if (nf >= za->n_entries || is_pending (za, nf)) {
					break;
				}
				z->grow = z->grow * 4u > WIN_MAX_ENTRIES ? WIN_MAX_ENTRIES : z->grow * 4u;
				struct window *nw = window_begin_ex (za, nf, z->grow, 0, 1);
				if (!nw) {
					break;
				}
				z->ahead[z->n_ahead++] = nw;
			}
		}
		if (window_wait (z->win) == 0) {
			w = z->win;
		} else if (z->win->quiet && attempt == 0) {
			/* a window prepared ahead did not make it (no free batch slot on the device, a layout the
			 * one-read path cannot take): start over at this entry, on this thread, and say why if it
			 * fails again */
			windows_drop (z);
		} else {
			return NULL;
		}
	}
	if (!w) {
		return NULL;
	}
	zip_uint64_t k = index - w->first;
	const struct otezip_entry *e = &za->entries[index];
	const ozg_result *r = &w->res[k];
	if (!w->valid[k]) {
		return NULL;
	}
	if (r->status == OZG_ST_BOMB) {
		fprintf (stderr, "mzip: entry '%s' claims huge uncompressed size (%u), rejecting to avoid zipbomb\n",
			e->name ? e->name : "<unknown>", e->uncomp_size); /* ref :459 */
		return NULL;
	}
	if (r->status != OZG_ST_OK) {
		return NULL;
	}
	if ((r->flags & OZG_F_REF_FALSE_REJECT) && getenv ("OTEZIP_REF_STRICT")) {
		return NULL; /* opt-in: behave like the reference's inflate on these streams */
	}
	if (r->flags & OZG_F_CRC_MISMATCH) {
		if (otezip_verify_crc) {
			return NULL;
		}
		fprintf (stderr, "Warning: CRC mismatch for '%s' (expected 0x%08x, got 0x%08x)\n", e->name ? e->name : "<unknown>",
			e->crc32, r->crc32); /* ref :676 */
	}
	struct ozh_file *f = (struct ozh_file *)calloc (1, sizeof (*f));
	if (!f) {
		return NULL;
	}
	f->pub.data = w->b.out + w->off[k];
	f->pub.size = e->uncomp_size;
	f->pub.pos = 0;
	f->win = w;
	w->readers++;
	return &f->pub;
}

int zip_fclose(zip_file_t *zf) {
	if (!zf) {
		return -1;
	}
	struct ozh_file *f = (struct ozh_file *)zf;
	if (f->win) {
		if (--f->win->readers == 0 && f->win->detached) {
			window_free (f->win);
		}
	} else {
		free (zf->data);
	}
	free (f);
	return 0;
}

zip_int64_t zip_fread(zip_file_t *zf, void *buf, zip_uint64_t nbytes) {
	if (!zf || !buf) {
		return -1;
	}
	if (zf->pos >= zf->size) {
		return 0;
	}
	zip_uint64_t left = zf->size - zf->pos, n = nbytes < left ? nbytes : left;
	memcpy (buf, zf->data + zf->pos, n);
	zf->pos += n;
	return (zip_int64_t)n;
}

/* --------------------------------------------------------------------- add */

zip_source_t *zip_source_buffer(zip_t *za, const void *data, zip_uint64_t len, int freep) {
	(void)za;
	zip_source_t *s = (zip_source_t *)malloc (sizeof (*s));
	if (!s) {
		return NULL;
	}
	s->buf = data;
	s->len = len;
	s->freep = freep;
	return s;
}

zip_source_t *zip_source_buffer_create(const void *data, zip_uint64_t len, int freep, zip_error_t *error) {
	(void)error;
	return zip_source_buffer (NULL, data, len, freep);
}

void zip_source_free(zip_source_t *src) {
	if (!src) {
		return;
	}
	if (src->freep && src->buf) {
		free ((void *)src->buf);
	}
	free (src);
}

/* otezip_compress_data (ref :788-1076) + the CRC of :1124 for the pending sources, on the GPU:
 * ONE ozg_encode_batch_host call returns every payload, the method each ended up with (the
 * reference's "not smaller -> STORE" rule is applied on the device) and the CRC-32 of each
 * input; local headers and payloads are then written in index order.  Methods 93/100/97
 * always end as STORE, exactly as in the reference, whose writers for them only add bytes
 * (SURVEY F9). */
/* Chunk buffers of the standard size are kept for the next flush (of this or another archive) instead of going
 * back to the allocator: a fresh 32 MiB block is page-faulted in while the caller's bytes are copied into it,
 * and that, not the copy, was two thirds of the time of a run of zip_file_add calls (tools/api_add_trace.sh).
 * At most CHUNK_POOL_MAX buffers (1 GiB) stay parked. */
#define CHUNK_POOL_MAX 32
static pthread_mutex_t g_chunk_mu = PTHREAD_MUTEX_INITIALIZER;
static uint8_t *g_chunk_pool[CHUNK_POOL_MAX];
static int g_chunk_n;

static uint8_t *chunk_alloc(size_t cap) {
	if (cap == PCHUNK_BYTES) {
		uint8_t *p = NULL;
		pthread_mutex_lock (&g_chunk_mu);
		if (g_chunk_n > 0) {
			p = g_chunk_pool[--g_chunk_n];
		}
		pthread_mutex_unlock (&g_chunk_mu);
		if (p) {
			return p;
		}
	}
	return (uint8_t *)malloc (cap);
}

static void chunk_free(uint8_t *p, size_t cap) {
	if (p && cap == PCHUNK_BYTES) {
		pthread_mutex_lock (&g_chunk_mu);
		if (g_chunk_n < CHUNK_POOL_MAX) {
			g_chunk_pool[g_chunk_n++] = p;
			p = NULL;
		}
		pthread_mutex_unlock (&g_chunk_mu);
	}
	free (p);
}

static void pending_drop(struct ozh_zip *z) {
	for (size_t k = 0; k < z->npch; k++) {
		chunk_free (z->pch[k].buf, z->pch[k].cap);
	}
	z->npch = 0;
	z->pn = 0;
	z->ptotal = 0;
}

/* one chunk's batch on its way through the GPU */
struct enc_flight {
	struct winbuf b;
	ozg_enc_entry *ee;
	ozg_job *job;
	size_t chunk;
	int live;
};

/* copy chunk k into pinned memory and submit its encode */
static int enc_start(zip_t *za, size_t k, struct enc_flight *f) {
	struct ozh_zip *z = (struct ozh_zip *)za;
	const struct pchunk *c = &z->pch[k];
	memset (f, 0, sizeof (*f));
	f->chunk = k;
	f->ee = (ozg_enc_entry *)calloc (c->nitems ? c->nitems : 1, sizeof (*f->ee));
	if (!f->ee) {
		return -1;
	}
	uint64_t ocap = 0;
	for (size_t i = 0; i < c->nitems; i++) {
		const struct pend_item *it = &z->pit[c->item0 + i];
		const struct otezip_entry *e = &za->entries[it->index];
		f->ee[i].src_off = it->off;
		f->ee[i].src_size = it->len;
		f->ee[i].method = e->method; /* (what zip_set_file_compression left there) */
		f->ee[i].dst_off = ocap;
		const uint64_t cap = ozg_compress_bound (e->method, it->len);
		f->ee[i].dst_cap = (uint32_t)cap;
		ocap += (cap + 15u) & ~15ull;
	}
	if (winbuf_get (z, &f->b, c->used + 64, (size_t)ocap + 64, c->nitems) != 0) {
		fprintf (stderr, "otezip-amd: %s\n", ozg_last_error ());
		free (f->ee);
		f->ee = NULL;
		return -1;
	}
	copy_span (f->b.src, c->buf, c->used);
	/* the devices take the batches in turn (OTEZIP_GPU_DEVICES), like the reader's windows */
	if (ozg_use_device (g_dev0 + (int) (k % (size_t) (g_ndev > 0 ? g_ndev : 1))) != OZG_OK ||
		ozg_encode_submit_packed (f->b.src, c->used + 16, f->b.out, ocap + 16, f->ee, (ozg_enc_result *)f->b.res, (uint32_t)c->nitems, &f->job) != OZG_OK) {
		fprintf (stderr, "otezip-amd: GPU encode failed: %s\n", ozg_last_error ());
		winbuf_put (&f->b);
		free (f->ee);
		f->ee = NULL;
		return -1;
	}
	f->live = 1;
	TRACE ("encode batch %zu: %zu entries, %zu bytes submitted", k, c->nitems, c->used);
	return 0;
}

/* wait for a chunk's batch and append its local headers + payloads to the file; *written counts
 * the pending items that are completely in the file, *at is the file offset behind the last of them */
static int enc_finish(zip_t *za, struct enc_flight *f, uint64_t *at, size_t *written) {
	struct ozh_zip *z = (struct ozh_zip *)za;
	int rc = -1;
	if (!f->live) {
		return -1;
	}
	const struct pchunk *c = &z->pch[f->chunk];
	const ozg_enc_result *er = (const ozg_enc_result *)f->b.res;
	if (ozg_job_wait (f->job) != OZG_OK) {
		fprintf (stderr, "otezip-amd: GPU encode failed: %s\n", ozg_last_error ());
		goto done;
	}
	uint64_t poff = 0; /* the payloads come back packed: each at the running 16-byte-aligned offset */
	for (size_t i = 0; i < c->nitems; i++) {
		struct otezip_entry *e = &za->entries[z->pit[c->item0 + i].index];
		if (er[i].status != OZG_ST_OK || *at > UINT32_MAX) {
			goto done;
		}
		const struct otezip_entry keep = *e;
		e->method = er[i].method;
		e->crc32 = er[i].crc32;
		e->comp_size = er[i].comp_size;
		e->local_hdr_ofs = (uint32_t)*at;
		const uint32_t hl = write_lfh (za->fp, e);
		if (ferror (za->fp) || (er[i].comp_size && fwrite (f->b.out + poff, 1, er[i].comp_size, za->fp) != er[i].comp_size)) {
			char *nm = e->name;
			*e = keep;
			e->name = nm;
			goto done;
		}
		*at += (uint64_t)hl + er[i].comp_size;
		poff += ((uint64_t)er[i].comp_size + 15u) & ~15ull;
		(*written)++;
	}
	rc = 0;
	TRACE ("encode batch %zu written", f->chunk);
done:
	winbuf_put (&f->b);
	free (f->ee);
	f->ee = NULL;
	f->live = 0;
	return rc;
}

/* Encode everything that is pending and write it: the next chunks are on the GPUs (one per device, in
 * turn) while chunk k's output goes to the file.  On failure the entries that did not reach the file
 * are taken out of the directory again (adds; zip_file_replace restores its entry itself): the
 * archive then lists exactly what it holds, as with the reference, which fails inside zip_file_add
 * and never records such an entry (ref otezip.c:1151-1183). */
static int flush_pending_ex(zip_t *za, int is_replace) {
	struct ozh_zip *z = (struct ozh_zip *)za;
	if (z->pn == 0) {
		return 0;
	}
	int rc = -1;
	struct enc_flight fl[FLIGHTS_MAX];
	memset (fl, 0, sizeof (fl));
	size_t written = 0;
	uint64_t at = 0;
	int have_at = 0;
	if (!gpu_up ()) {
		goto done;
	}
	ozg_use_device (g_dev0);
	if (fseek (za->fp, 0, SEEK_END) != 0) {
		goto done;
	}
	long pos = ftell (za->fp);
	if (pos < 0) {
		goto done;
	}
	at = (uint64_t)pos;
	have_at = 1;
	const size_t nfl = (size_t) (g_ndev > 1 ? g_ndev + 1 : 2);
	for (size_t k = 0; k < z->npch + nfl - 1u; k++) {
		if (k < z->npch && enc_start (za, k, &fl[k % nfl]) != 0) {
			goto done;
		}
		if (k + 1u >= nfl && enc_finish (za, &fl[(k + 1u - nfl) % nfl], &at, &written) != 0) {
			goto done;
		}
	}
	rc = 0;
done:
	ozg_use_device (g_dev0);
	for (size_t k = 0; k < FLIGHTS_MAX; k++) {
		if (fl[k].live) {
			(void)ozg_job_wait (fl[k].job);
			winbuf_put (&fl[k].b);
			free (fl[k].ee);
		}
	}
	if (rc != 0) {
		z->write_failed = 1;
		if (have_at) { /* nothing behind the last complete entry stays in the file */
			fflush (za->fp);
			if (ftruncate (fileno (za->fp), (off_t)at) != 0) {
				/* (the directory is written behind whatever is there: still a valid archive) */
			}
			fseek (za->fp, 0, SEEK_END);
		}
		if (!is_replace && written < z->pn) {
			const zip_uint64_t keep = z->pit[written].index; /* adds only append: the unwritten ones are the tail */
			for (zip_uint64_t i = keep; i < za->n_entries; i++) {
				free (za->entries[i].name);
				za->entries[i].name = NULL;
			}
			za->n_entries = keep;
			za->next_index = keep;
		}
	}
	pending_drop (z);
	return rc;
}

static int flush_pending(zip_t *za) { return flush_pending_ex (za, 0); }

/* room for len more source bytes and one more item; returns where the bytes go */
static uint8_t *pending_reserve(struct ozh_zip *z, size_t len, uint64_t *off, int exact) {
	const size_t alen = (len + 15u) & ~(size_t)15u;
	if (z->pn == z->pncap) {
		size_t nc = z->pncap ? 2 * z->pncap : 1024;
		struct pend_item *ni = (struct pend_item *)realloc (z->pit, nc * sizeof (*ni));
		if (!ni) {
			return NULL;
		}
		z->pit = ni;
		z->pncap = nc;
	}
	struct pchunk *c = z->npch ? &z->pch[z->npch - 1] : NULL;
	if (!c || c->used + alen + 64 > c->cap || c->nitems >= PCHUNK_ITEMS) {
		if (z->npch == z->pchcap) {
			size_t nc = z->pchcap ? 2 * z->pchcap : 16;
			struct pchunk *np = (struct pchunk *)realloc (z->pch, nc * sizeof (*np));
			if (!np) {
				return NULL;
			}
			z->pch = np;
			z->pchcap = nc;
		}
		c = &z->pch[z->npch];
		memset (c, 0, sizeof (*c));
		c->cap = exact || alen + 64 > PCHUNK_BYTES ? alen + 64 : PCHUNK_BYTES; /* (a replace is a batch of one) */
		c->buf = chunk_alloc (c->cap);
		if (!c->buf) {
			return NULL;
		}
		c->item0 = z->pn;
		z->npch++;
	}
	*off = c->used;
	uint8_t *at = c->buf + c->used;
	c->used += alen;
	c->nitems++;
	z->ptotal += alen;
	return at;
}

zip_int64_t zip_file_add(zip_t *za, const char *name, zip_source_t *src, zip_flags_t flags) {
	/* ref otezip.c:1079-1183 */
	(void)flags;
	if (!za || !name || !src || za->mode != 1) {
		return -1;
	}
	struct ozh_zip *z = (struct ozh_zip *)za;
	size_t nl = strlen (name);
	if (nl > MAX_FIELD_LEN || src->len > MAX_PAYLOAD || src->len > UINT32_MAX) {
		return -1;
	}
	if (g_gpu_state < 0) {
		return -1; /* there is no CPU encoder to fall back to */
	}
	gpu_up_in_background ();
	if (z->pn >= PEND_FLUSH_ITEMS || (z->pn && z->ptotal + src->len > PEND_FLUSH_BYTES)) {
		if (flush_pending (za) != 0) {
			return -1;
		}
	}
	/* everything that can fail comes before the reservation in the pending chunk */
	char *nm = (char *)malloc (nl + 1);
	if (!nm) {
		return -1;
	}
	struct otezip_entry *ne = (struct otezip_entry *)realloc (za->entries, (za->n_entries + 1) * sizeof (*ne));
	if (!ne) {
		free (nm);
		return -1;
	}
	za->entries = ne;
	struct otezip_entry *e = &za->entries[za->n_entries];
	memset (e, 0, sizeof (*e));
	uint64_t poff = 0;
	uint8_t *pat = pending_reserve (z, (size_t)src->len, &poff, 0);
	if (!pat) {
		free (nm);
		return -1; /* src stays with the caller, as in the reference */
	}
	e->name = nm;
	memcpy (e->name, name, nl + 1);
	e->method = za->default_method > 0 ? za->default_method : 0;
	e->uncomp_size = (uint32_t)src->len;
	dostime_now (&e->file_time, &e->file_date);
	e->external_attr = 0100644u << 16;
	/* the bytes are copied now: like the reference, the call is done with the caller's buffer
	 * when it returns (comp_size, CRC and offset follow when the batch is encoded) */
	struct pend_item *it = &z->pit[z->pn++];
	it->index = za->n_entries;
	it->off = poff;
	it->len = (uint32_t)src->len;
	if (src->len) {
		memcpy (pat, src->buf, (size_t)src->len);
	}
	if (src->freep) {
		free ((void *)src->buf);
	}
	free (src);
	zip_uint64_t index = za->n_entries++;
	za->next_index = za->n_entries;
	return (zip_int64_t)index;
}

int zip_file_replace(zip_t *za, zip_uint64_t index, zip_source_t *src, zip_flags_t flags) {
	/* ref otezip.c:1617-1658: a new local header + payload at the end of the file */
	(void)flags;
	if (!valid (za) || index >= za->n_entries || !src || za->mode != 1 || src->len > UINT32_MAX || src->len > MAX_PAYLOAD) {
		return -1;
	}
	struct ozh_zip *z = (struct ozh_zip *)za;
	if (!gpu_up () || flush_pending (za) != 0) {
		return -1;
	}
	/* the entry's new bytes: a batch of one, written at once */
	uint64_t poff = 0;
	uint8_t *pat = pending_reserve (z, (size_t)src->len, &poff, 1);
	if (!pat) {
		return -1;
	}
	struct otezip_entry *e = &za->entries[index];
	const struct otezip_entry keep = *e;
	e->uncomp_size = (uint32_t)src->len; /* (the entry keeps its method, ref :1630) */
	struct pend_item *it = &z->pit[z->pn++];
	it->index = index;
	it->off = poff;
	it->len = (uint32_t)src->len;
	if (src->len) {
		memcpy (pat, src->buf, (size_t)src->len);
	}
	if (flush_pending_ex (za, 1) != 0) {
		char *nm = e->name;
		*e = keep;
		e->name = nm;
		return -1;
	}
	windows_drop (z); /* decoded copies of the old bytes must not be served */
	return 0;
}

int zip_replace(zip_t *za, zip_uint64_t index, zip_source_t *src) { return zip_file_replace (za, index, src, 0); }
zip_int64_t zip_add(zip_t *za, const char *name, zip_source_t *src) { return zip_file_add (za, name, src, 0); }

int zip_set_file_compression(zip_t *za, zip_uint64_t index, zip_int32_t comp, zip_uint32_t comp_flags) {
	/* ref otezip.c:1186-1237 validates the method and then relabels the entry even though its
	 * bytes are already on disk, which makes the archive unreadable (the
	 * reference's own test_set_file_compression.c fails on it).  Here an added entry is
	 * encoded when its batch is flushed, so until then the call does what libzip's does: it
	 * selects the method the entry is written with.  For an entry that is already on disk
	 * the label keeps describing the bytes; the call validates and succeeds. */
	(void)comp_flags;
	if (!valid (za) || index >= za->n_entries || za->mode != 1) {
		return -1;
	}
	switch (comp) {
	case OTEZIP_METHOD_STORE:
	case OTEZIP_METHOD_DEFLATE:
	case OTEZIP_METHOD_ZSTD:
	case OTEZIP_METHOD_LZFSE:
	case OTEZIP_METHOD_LZMA:
	case OTEZIP_METHOD_BROTLI:
		break;
	default:
		return -1;
	}
	if (is_pending (za, index)) {
		za->entries[index].method = (uint16_t)comp;
	}
	return 0;
}
