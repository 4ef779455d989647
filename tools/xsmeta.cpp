// xsmeta -- metafile tools on the C ABI (no GPU needed):
//
//   xsmeta cat FILE.meta [--mapping-data]
//       human-readable dump, same fields and layout as the reference's
//       metafile_cat.cpp:23-52 (compression type, per chunk: original/actual
//       byte offset and size, number of 'byte offset -> line index' pairs)
//   xsmeta write INPUT --meta OUT.meta [--data OUT] [--none|--lz4|--lz4hc|--zstd]
//                [--chunk-bytes N] [--gap N]
//       preprocess INPUT into the reference's on-disk layout (what produced
//       test/files/sample.*.meta): newline-aligned chunks, mapping entries every
//       >= gap bytes at a line start, chunk-wise compression
#include <xsg.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

static const char* type_name(int32_t t) {
  switch (t) {
    case XSG_COMPRESSION_NONE: return "NONE";
    case XSG_COMPRESSION_ZSTD: return "ZSTD";
    case XSG_COMPRESSION_LZ4: return "LZ4";
    default: return "UNKNOWN";
  }
}

static int cat(const char* path, bool mapping) {
  int32_t comp = 0;
  xsg_file_chunk* chunks = nullptr;
  uint64_t n = 0, npairs = 0;
  uint64_t* maps = nullptr;
  if (xsg_meta_read(path, &comp, &chunks, &n, mapping ? &maps : nullptr, mapping ? &npairs : nullptr) != XSG_OK) {
    std::fprintf(stderr, "xsmeta: %s\n", xsg_last_error());
    return 1;
  }
  std::printf("Compression Type: %s (%zu)\n", type_name(comp), sizeof(comp));
  uint64_t at = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const xsg_file_chunk& c = chunks[i];
    std::printf("---\nChunk %llu:\n Byte offset:\n  original: %llu\n  actual  : %llu\n Size (bytes):\n  original: %llu\n"
                "  actual  : %llu\n 'Byte offset -> line index' mapping data (total: %llu):\n",
                (unsigned long long)i, (unsigned long long)c.original_offset, (unsigned long long)c.actual_offset,
                (unsigned long long)c.original_size, (unsigned long long)c.actual_size,
                (unsigned long long)c.n_mappings);
    if (mapping) {
      for (uint64_t k = 0; k < c.n_mappings; ++k, ++at)
        std::printf("  global byte offset: %llu\n  global line index : %llu\n  ---\n", (unsigned long long)maps[2 * at],
                    (unsigned long long)maps[2 * at + 1]);
    }
  }
  xsg_free(chunks);
  xsg_free(maps);
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 3 && std::strcmp(argv[1], "cat") == 0) {
    return cat(argv[2], argc >= 4 && std::strcmp(argv[3], "--mapping-data") == 0);
  }
  if (argc >= 3 && std::strcmp(argv[1], "write") == 0) {
    std::string input = argv[2], meta, data;
    int32_t comp = XSG_COMPRESSION_NONE;
    int hc = 0;
    uint64_t chunk = 16u << 20, gap = 500;
    for (int i = 3; i < argc; ++i) {
      const std::string a = argv[i];
      if (a == "--meta" && i + 1 < argc) meta = argv[++i];
      else if (a == "--data" && i + 1 < argc) data = argv[++i];
      else if (a == "--none") comp = XSG_COMPRESSION_NONE;
      else if (a == "--lz4") comp = XSG_COMPRESSION_LZ4;
      else if (a == "--lz4hc") { comp = XSG_COMPRESSION_LZ4; hc = 1; }
      else if (a == "--zstd") comp = XSG_COMPRESSION_ZSTD;
      else if (a == "--chunk-bytes" && i + 1 < argc) chunk = std::strtoull(argv[++i], nullptr, 10);
      else if (a == "--gap" && i + 1 < argc) gap = std::strtoull(argv[++i], nullptr, 10);
      else { std::fprintf(stderr, "xsmeta: unknown argument '%s'\n", a.c_str()); return 2; }
    }
    if (meta.empty() || (comp != XSG_COMPRESSION_NONE && data.empty())) {
      std::fprintf(stderr, "xsmeta write: --meta (and --data for compressed output) required\n");
      return 2;
    }
    if (xsg_meta_write(input.c_str(), meta.c_str(), data.empty() ? nullptr : data.c_str(), comp, chunk, gap, hc) != XSG_OK) {
      std::fprintf(stderr, "xsmeta: %s\n", xsg_last_error());
      return 1;
    }
    return 0;
  }
  std::fprintf(stderr, "usage: %s cat FILE.meta [--mapping-data]\n       %s write INPUT --meta OUT.meta [--data OUT] "
                       "[--none|--lz4|--lz4hc|--zstd] [--chunk-bytes N] [--gap N]\n", argv[0], argv[0]);
  return 2;
}
