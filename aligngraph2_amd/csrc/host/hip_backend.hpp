// The product's only graph backend: the HIP library behind include/pagraph_hip.h.
#pragma once
#include <memory>

#include "pagraph_driver.hpp"

namespace pagh {
std::unique_ptr<GraphBackend> makeHipBackend(int deviceOrdinal);
}
