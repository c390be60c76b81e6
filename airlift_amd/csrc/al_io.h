// al_io.h -- host sequence-file reader + base tables (product code)
#pragma once
#include <string>
struct AlSeqFile;
AlSeqFile *al_sf_open(const char *fn);
void al_sf_close(AlSeqFile *f);
// returns sequence length, -1 at EOF.  qual empty for FASTA.
int al_sf_read(AlSeqFile *f, std::string &name, std::string &seq, std::string &qual);
const unsigned char *al_nt4();
const unsigned char *al_comp();
