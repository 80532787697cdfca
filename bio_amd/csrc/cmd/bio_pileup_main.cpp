// bio-pileup executable: the work-alike of cmd/bio-pileup/main.go over libbiopileup.so.
#include "../host/bio_host.h"

int main(int argc, char** argv) { return bio::BioPileupMain(argc, argv); }
