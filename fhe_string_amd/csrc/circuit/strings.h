// strings.h -- compile a string operation of the reference into a level-scheduled program.
#pragma once
#include <string>
#include <vector>

#include "radix.h"

namespace fhes {
namespace circ {

// op: see include/fhe_string_amd.h (fhes_program_compile); shape: slot count of each string
// operand (chars incl. terminator and padding).  Throws std::invalid_argument / logic_error.
// operand signature of an operation ("SS", "SUS", ...: S string, U usize, I clear int) or nullptr
const char *op_signature(const std::string &op);

Program compile_op(const fhes_params &p, const std::string &op, const std::vector<int32_t> &shape,
                   int32_t num_blocks_usize);

}  // namespace circ
}  // namespace fhes
