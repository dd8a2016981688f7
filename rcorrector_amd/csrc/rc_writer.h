// rc_writer.h -- output side of the `rcorrector` CLI: the formatted slices of a batch written in input order
// (Reads.h:360-421 prints record by record through one FILE / gzFile per input file).
#pragma once
#include "rc_format.h"

// the slices of a batch, in order, behind what the file holds already
void emit_slices(ReadFile &f, const std::vector<OutBuf> &sl);
