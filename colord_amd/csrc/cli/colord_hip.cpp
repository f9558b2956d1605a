// colord_hip — command-line tool on top of libcolord_hip.so with the reference's sub-commands (src/colord/main.cpp,
// arg_parse.cpp): compress-ont | compress-pbhifi | compress-pbraw (compress.cpp: GPU data path), decompress and info
// (decompress.cpp: host decoders of the library).  Archives are interchangeable with the reference's in both directions.
#include <cstdio>
#include <cstdlib>
#include <string>

int run_compress(int argc, char** argv);        // compress.cpp
int run_decompress(int argc, char** argv);      // decompress.cpp
int run_info(int argc, char** argv);
int run_parse_check(int argc, char** argv);     // compress.cpp: the input reader alone (test aid, no GPU)
int run_rccl_selftest(int argc, char** argv);   // compress.cpp: the collectives of the multi-GPU host over RCCL

int main(int argc, char** argv)
{
	setenv("GPU_MAX_HW_QUEUES", "32", 0);            // before the HIP runtime starts: the compressor's contexts keep more than 4 streams busy
	const std::string cmd = argc >= 2 ? argv[1] : "";
	if (cmd == "decompress") return run_decompress(argc, argv);
	if (cmd == "info") return run_info(argc, argv);
	if (cmd == "parse-check") return run_parse_check(argc, argv);
	if (cmd == "rccl-selftest") return run_rccl_selftest(argc, argv);
	if (argc < 2 || cmd == "-h" || cmd == "--help")
	{
		const char* a[] = { argv[0], "compress-ont", "--help" };
		return run_compress(3, (char**)a);
	}
	return run_compress(argc, argv);
}
