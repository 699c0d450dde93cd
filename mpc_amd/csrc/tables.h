// filled in by the table-construction kernels
#pragma once
