// Load a (possibly compressed / archived) text file into memory for the Matrix-Market reader.
//
// Reference behaviour being replaced: lib/parallel_io.c:28-130 (file_load with auto_decompress): extensions are peeled
// right to left — "gz"/"zst" -> `zstdmt -d --stdout`, "tar" -> `tar -x -O` — through a shell pipeline into a temporary
// file under /tmp, which is then mmap'ed. Here the same extensions are handled IN PROCESS, with no shell, no temporary
// file and no second pass over the disk: gzip members are inflated with zlib straight into the parse buffer, zstd frames
// through libzstd.so.1 (bound at run time: the image ships the library but not its header), and a tar stream yields its
// first regular member (SuiteSparse archives are <name>/<name>.mtx first; the reference concatenates ALL members, which
// only parses when there is exactly one).

#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <string>

#include "host.hpp"

namespace spmv_host {

void
FileBuf::release()
{
	if (mapped && data)
		munmap((void *) data, size);
	else if (data)
		free((void *) data);
	data = nullptr;
	size = 0;
	mapped = false;
}

static bool
ends_with(const std::string & s, const char * ext)
{
	const size_t n = strlen(ext);
	return s.size() > n && s.compare(s.size() - n, n, ext) == 0;
}

// all gzip members of [src, src+n) -> malloc'ed buffer
static int
gunzip(const char * path, const unsigned char * src, size_t n, FileBuf & out)
{
	// ISIZE (last 4 bytes, mod 2^32) is a good first guess for single-member files
	size_t cap = n >= 4 ? (size_t) src[n - 4] | (size_t) src[n - 3] << 8 | (size_t) src[n - 2] << 16 | (size_t) src[n - 1] << 24 : 0;
	if (cap < n)
		cap = n * 4;
	cap += 64;
	unsigned char * dst = (unsigned char *) malloc(cap);
	if (!dst)
	{
		set_error("out of memory inflating '%s'", path);
		return 1;
	}
	z_stream zs;
	memset(&zs, 0, sizeof(zs));
	if (inflateInit2(&zs, 15 + 32) != Z_OK)                         // gzip or zlib header, auto-detected
	{
		free(dst);
		set_error("zlib: inflateInit2 failed for '%s'", path);
		return 1;
	}
	size_t in_pos = 0, out_pos = 0;
	int rc = Z_OK;
	bool member_complete = false;                                   // the input must END at the end of a member
	while (in_pos < n)
	{
		if (out_pos == cap)
		{
			cap = cap * 2;
			unsigned char * d2 = (unsigned char *) realloc(dst, cap);
			if (!d2)
			{
				inflateEnd(&zs);
				free(dst);
				set_error("out of memory inflating '%s'", path);
				return 1;
			}
			dst = d2;
		}
		const size_t in_chunk = std::min<size_t>(n - in_pos, 1u << 30), out_chunk = std::min<size_t>(cap - out_pos, 1u << 30);
		zs.next_in = const_cast<unsigned char *>(src + in_pos);
		zs.avail_in = (uInt) in_chunk;
		zs.next_out = dst + out_pos;
		zs.avail_out = (uInt) out_chunk;
		rc = inflate(&zs, Z_NO_FLUSH);
		member_complete = rc == Z_STREAM_END;
		in_pos += in_chunk - zs.avail_in;
		out_pos += out_chunk - zs.avail_out;
		if (rc == Z_STREAM_END)
		{
			if (in_pos < n && inflateReset(&zs) != Z_OK)                // next member of a multi-member file
				break;
			rc = Z_OK;
			continue;
		}
		if (rc != Z_OK && rc != Z_BUF_ERROR)
			break;
		if (rc == Z_BUF_ERROR && zs.avail_in == 0 && zs.avail_out != 0)
			break;                                                      // truncated input
	}
	inflateEnd(&zs);
	if (rc != Z_OK || in_pos < n || !member_complete)
	{
		free(dst);
		set_error("'%s': corrupt or truncated gzip data (zlib rc %d)", path, rc);
		return 1;
	}
	out.release();
	out.data = (const char *) dst;
	out.size = out_pos;
	out.mapped = false;
	return 0;
}

// libzstd's stable one-shot API, bound at run time
static int
unzstd(const char * path, const unsigned char * src, size_t n, FileBuf & out)
{
	typedef size_t (*decompress_t)(void *, size_t, const void *, size_t);
	typedef unsigned long long (*content_size_t)(const void *, size_t);
	typedef size_t (*frame_size_t)(const void *, size_t);
	typedef unsigned (*is_error_t)(size_t);
	static void * lib = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
	if (!lib)
	{
		set_error("'%s': zstd input needs libzstd.so.1 (%s)", path, dlerror());
		return 1;
	}
	decompress_t decompress = (decompress_t) dlsym(lib, "ZSTD_decompress");
	content_size_t content_size = (content_size_t) dlsym(lib, "ZSTD_getFrameContentSize");
	frame_size_t frame_size = (frame_size_t) dlsym(lib, "ZSTD_findFrameCompressedSize");
	is_error_t is_error = (is_error_t) dlsym(lib, "ZSTD_isError");
	if (!decompress || !content_size || !frame_size || !is_error)
	{
		set_error("'%s': libzstd.so.1 lacks the one-shot API", path);
		return 1;
	}
	// frames are independent: size them first, then decompress each into its place
	size_t total = 0;
	for (size_t pos = 0; pos < n;)
	{
		const unsigned long long cs = content_size(src + pos, n - pos);
		const size_t fs = frame_size(src + pos, n - pos);
		if (cs == (unsigned long long) -1 || cs == (unsigned long long) -2 || is_error(fs))
		{
			set_error("'%s': zstd frame without a content size or corrupt frame (streaming-compressed input is not supported)", path);
			return 1;
		}
		total += (size_t) cs;
		pos += fs;
	}
	unsigned char * dst = (unsigned char *) malloc(total + 64);
	if (!dst)
	{
		set_error("out of memory decompressing '%s'", path);
		return 1;
	}
	size_t out_pos = 0;
	for (size_t pos = 0; pos < n;)
	{
		const size_t fs = frame_size(src + pos, n - pos);
		const size_t got = decompress(dst + out_pos, total - out_pos, src + pos, fs);
		if (is_error(got))
		{
			free(dst);
			set_error("'%s': zstd decompression failed", path);
			return 1;
		}
		out_pos += got;
		pos += fs;
	}
	out.release();
	out.data = (const char *) dst;
	out.size = out_pos;
	out.mapped = false;
	return 0;
}

// first regular member of a tar stream (ustar / GNU headers; long names via 'L' records are skipped over)
static int
untar_first(const char * path, FileBuf & io)
{
	const char * p = io.data, * e = io.data + io.size;
	while (p + 512 <= e)
	{
		bool zero = true;
		for (int i = 0; i < 512 && zero; i++)
			zero = p[i] == 0;
		if (zero)
			break;
		char szbuf[13];
		memcpy(szbuf, p + 124, 12);
		szbuf[12] = 0;
		const size_t sz = (size_t) strtoull(szbuf, NULL, 8);
		const char type = p[156];
		const char * body = p + 512;
		if (sz > (size_t) (e - body))              // compared as sizes: `body + sz` would wrap for a crafted octal size
			break;
		if ((type == '0' || type == 0) && sz > 0)
		{
			char * copy = (char *) malloc(sz + 64);
			if (!copy)
			{
				set_error("out of memory extracting '%s'", path);
				return 1;
			}
			memcpy(copy, body, sz);
			io.release();
			io.data = copy;
			io.size = sz;
			io.mapped = false;
			return 0;
		}
		p = body + (sz + 511) / 512 * 512;
	}
	set_error("'%s': no regular file inside the tar archive", path);
	return 1;
}

int
file_load(const char * path, FileBuf & out)
{
	out.release();
	int fd = open(path, O_RDONLY);
	if (fd < 0)
	{
		set_error("cannot open '%s': %s", path, strerror(errno));
		return 1;
	}
	struct stat st;
	if (fstat(fd, &st) || !S_ISREG(st.st_mode))
	{
		close(fd);
		set_error("'%s': not a file", path);                            // parallel_io.c:118-119
		return 1;
	}
	const size_t N = (size_t) st.st_size;
	if (N == 0)
	{
		close(fd);
		return 0;                                                       // empty: the caller reports it
	}
	const char * buf = (const char *) mmap(NULL, N, PROT_READ, MAP_PRIVATE, fd, 0);
	close(fd);
	if (buf == MAP_FAILED)
	{
		set_error("mmap of '%s' failed: %s", path, strerror(errno));
		return 1;
	}
	out.data = buf;
	out.size = N;
	out.mapped = true;
	// peel extensions right to left, like the reference's loop (parallel_io.c:82-99)
	std::string name(path);
	while (true)
	{
		if (ends_with(name, ".gz") || ends_with(name, ".zst"))
		{
			const bool gz = ends_with(name, ".gz");
			name.resize(name.size() - (gz ? 3 : 4));
			FileBuf next;
			// the reference pipes both through zstdmt, which also reads gzip: decide by magic, not by name
			const unsigned char * s = (const unsigned char *) out.data;
			const bool gzip_magic = out.size >= 2 && s[0] == 0x1f && s[1] == 0x8b;
			if (gzip_magic ? gunzip(path, s, out.size, next) : unzstd(path, s, out.size, next))
			{
				out.release();
				return 1;
			}
			out.release();
			out = next;
			next.data = nullptr;
			continue;
		}
		if (ends_with(name, ".tar"))
		{
			name.resize(name.size() - 4);
			if (untar_first(path, out))
			{
				out.release();
				return 1;
			}
			continue;
		}
		if (ends_with(name, ".tgz"))
		{
			name.resize(name.size() - 4);
			name += ".tar.gz";
			continue;
		}
		break;
	}
	return 0;
}

}  // namespace spmv_host
