// A reader written against include/colord_api.h (the reference's API surface): archive info to stderr, records to stdout in
// the format of the command-line decompressor, so that the CPU suite can compare both with the reference's own output.
#include "colord_api.h"
#include <iostream>
int main(int argc, char** argv)
{
	if (argc < 2) { std::cerr << "usage: api_dump archive.colord [reference_genome]\n"; return 2; }
	try
	{
		colord::DecompressionStream stream(argv[1], argc > 2 ? argv[2] : "");
		const colord::Info info = stream.GetInfo();
		info.ToOstream(std::cerr);
		std::cerr << "total bases: " << info.totalBases << "\n";
		uint64_t n = 0;
		while (auto rec = stream.NextRecord())
		{
			if (info.isFastq) std::cout << "@" << rec.ReadHeader() << "\n" << rec.Read() << "\n+" << rec.QualHeader() << "\n" << rec.Qual() << "\n";
			else std::cout << ">" << rec.ReadHeader() << "\n" << rec.Read() << "\n";
			++n;
		}
		std::cerr << "records: " << n << "\n";
	}
	catch (const std::exception& e) { std::cerr << "Error: " << e.what() << "\n"; return 1; }
	return 0;
}
