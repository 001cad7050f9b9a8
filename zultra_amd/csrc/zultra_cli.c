/* zultra_cli.c — command-line compressor on top of libzultra_amd.so (include/libzultra.h).
 *
 * Mirrors the compress path of the reference's tool (tool/zultra.c:97-237: open, zultra_stream_init, feed, drain, finalize)
 * with the two things that path cannot do: it takes the max-block size (-b; the reference fixes it at the
 * 1 MiB default, tool/zultra.c:151) and it feeds the stream in MiB-sized chunks instead of 16 KiB ones (tool/zultra.c:98,161),
 * so that the device sees batches of many max-blocks. The bytes written are those of the reference for the same flags and
 * block size.
 *
 *    zultra_amd_cli [-b <max block size>] [-f gzip|zlib|raw] [-k <chunk KiB>] [-d <device>] [-v] <infile> <outfile>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/libzultra.h"

static double now_s(void) {
   struct timespec ts;
   clock_gettime(CLOCK_MONOTONIC, &ts);
   return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char **argv) {
   unsigned flags = ZULTRA_FLAG_GZIP_FRAMING, block = 0;
   size_t chunk = (size_t)8 << 20;
   int verbose = 0, i = 1;
   for (; i < argc && argv[i][0] == '-' && argv[i][1]; i++) {
      if (!strcmp(argv[i], "-b") && i + 1 < argc)
         block = (unsigned)strtoul(argv[++i], NULL, 0);
      else if (!strcmp(argv[i], "-k") && i + 1 < argc)
         chunk = (size_t)strtoul(argv[++i], NULL, 0) << 10;
      else if (!strcmp(argv[i], "-d") && i + 1 < argc)
         zultra_set_device(atoi(argv[++i]));
      else if (!strcmp(argv[i], "-f") && i + 1 < argc) {
         const char *f = argv[++i];
         flags = !strcmp(f, "gzip") ? ZULTRA_FLAG_GZIP_FRAMING : !strcmp(f, "zlib") ? ZULTRA_FLAG_ZLIB_FRAMING : ZULTRA_FLAG_DEFLATE_FRAMING;
      }
      else if (!strcmp(argv[i], "-v"))
         verbose = 1;
      else
         break;
   }
   if (argc - i != 2 || chunk == 0) {
      fprintf(stderr, "usage: %s [-b <max block size>] [-f gzip|zlib|raw] [-k <chunk KiB>] [-d <device>] [-v] <infile> <outfile>\n", argv[0]);
      return 100;
   }
   FILE *fin = fopen(argv[i], "rb");
   if (!fin) {
      fprintf(stderr, "error opening '%s' for reading\n", argv[i]);
      return 100;
   }
   FILE *fout = fopen(argv[i + 1], "wb");
   if (!fout) {
      fprintf(stderr, "error opening '%s' for writing\n", argv[i + 1]);
      fclose(fin);
      return 100;
   }
   unsigned char *in = (unsigned char *)malloc(chunk), *out = (unsigned char *)malloc(chunk);
   zultra_stream_t strm;
   memset(&strm, 0, sizeof(strm));
   if (!in || !out || zultra_stream_init(&strm, flags, block) != ZULTRA_OK) {
      fprintf(stderr, "error initializing compressor (no HIP device? this library has no CPU path)\n");
      return 100;
   }
   const double t0 = now_s();
   int status = ZULTRA_OK, eof = 0, rc = 0;
   while (status == ZULTRA_OK) {
      if (!strm.avail_in && !eof) {
         strm.next_in = in;
         strm.avail_in = fread(in, 1, chunk, fin);
         eof = strm.avail_in < chunk;   /* as the reference's tool: a short read ends the input (tool/zultra.c:161-166) */
      }
      strm.next_out = out;
      strm.avail_out = chunk;
      status = zultra_stream_compress(&strm, eof ? ZULTRA_FINALIZE : ZULTRA_CONTINUE);
      const size_t produced = chunk - strm.avail_out;
      if (produced && fwrite(out, 1, produced, fout) != produced) {
         fprintf(stderr, "write error\n");
         rc = 100;
         break;
      }
   }
   if (status != ZULTRA_STREAM_END && !rc) {
      fprintf(stderr, "compression error %d\n", status);
      rc = 100;
   }
   const double dt = now_s() - t0;
   if (verbose && !rc)
      fprintf(stdout, "%llu -> %llu bytes (%.2f %%), %.1f MB/s\n", (unsigned long long)strm.total_in, (unsigned long long)strm.total_out,
              strm.total_in ? 100.0 * (double)strm.total_out / (double)strm.total_in : 0.0, dt > 0 ? (double)strm.total_in / dt / 1e6 : 0.0);
   zultra_stream_end(&strm);
   zultra_release_cached_contexts();
   free(in);
   free(out);
   fclose(fin);
   fclose(fout);
   return rc;
}
